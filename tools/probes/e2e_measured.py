"""The bf16 e2e test's measured distances to the reference fixture (tests/test_e2e_gpu.py prints them), default path and with the pair
fusion off: the numbers its tolerances (1.5 x measured) are derived from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import test_e2e_gpu as T
from bonai_amd.debug import DBG
LOOSE = dict(feat=10.0, loss={k: 10.0 for k in T.TOL_BF16['loss']}, gradnorm=10.0, gradhead=10.0)
for off in (False, True, False):
    with DBG.override(no_pair_fusion=off):
        print('no_pair_fusion =', off)
        T._e2e_vs_fixture(LOOSE)
