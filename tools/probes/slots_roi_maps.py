# A/B leg: split-K slots (plain stores, summed by the batched unpack) only for the RoI-map weight gradients (FOA / mask heads),
# whose 252 workgroups each end in 65 536 fp32 atomics (29 % of the launch in the serialised profile)
from bonai_amd import kernels as K
K.WGRAD_SLOTS = lambda groups, B, OH, OW, Cin, Cout, T, ss, gos: T > 1 and B >= 128 and OH * OW <= 1024
