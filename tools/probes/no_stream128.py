# A/B leg: the 128-pixel stream tile off (launches with too few 256-pixel tiles back on the lock-step 128 x 128 kernel)
from bonai_amd import kernels as K


def _v(groups, B, OH, OW, Cin, Cout, T, ss, os_):
    M = B * OH * OW
    big = -(-M // 256) * (Cout // 256) * groups if Cout % 256 == 0 else 0
    half = -(-M // 128) * (Cout // 256) * groups if Cout % 256 == 0 else 0
    if Cout % 256 == 0 and big < 192 and half >= 192 and T * Cin >= 1024:
        return K.CONV_T128_FAST
    return K.CONV_AUTO


K.CONV_VARIANT = _v
