from bonai_amd import kernels as K
def _old(groups, B, OH, OW, Cin, Cout, T, ss, gos):
    narrow = Cin % 128 or Cout % 128
    big = Cin % 256 == 0 and Cout % 256 == 0 and B * OH * OW * (Cout // 256) * (Cin // 256) * T * groups >= 524288
    return K.WGRAD_AUTO if (narrow or big) else K.WGRAD_T128
K.WGRAD_VARIANT = _old
