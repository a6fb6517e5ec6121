import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# Collection order: the driver runs `pytest -m gpu -x`, so ONE failing composite test hides every
# test collected after it.  Oracle-parity files of single kernels come first, whole-model parity next, and the composite
# trainer / lifetime / multi-process tests -- the ones with the most moving parts -- last.  Files not named here keep their
# alphabetical place between the two groups.
_FIRST = ['test_lib_symbols.py', 'test_roi_nms_gpu.py', 'test_glue_gpu.py', 'test_offset_head_gpu.py', 'test_inference_gpu.py',
          'test_conv_gpu.py', 'test_conv_variants_gpu.py', 'test_fp16_gpu.py', 'test_data_gpu.py', 'test_deform_gpu.py',
          'test_hrnet_gpu.py', 'test_fullsize_props_gpu.py', 'test_e2e_gpu.py']
_LAST = ['test_lifetime_gpu.py', 'test_edge_gpu.py', 'test_trainer_gpu.py', 'test_ddp_gpu.py']


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)          # stable: the order inside a file (and among unnamed files) is pytest's own
    if os.environ.get('LOFT_TEST_ORDER') == 'reverse':      # (the suite must not depend on its file order)
        items.reverse()


@pytest.fixture(params=['planes_f16', 'planes_f16x4', 'planes_bf16', 'split6', 'split3', 'exact'])
def f32_contract(request):
    """The contractions of the fp32 parity mode (include/loft_hip.h): the mode's default since round 5, binary16 operand PLANES on the
    software-pipelined stream kernels (two planes per fp32 tensor under a power-of-two scale, three products: 22 significant bits),
    the binary16 planes with the lo x lo product as a fourth term, three bfloat16 planes / six products (24 bits), and the kernels of
    rounds 1-4: SPLIT6 (three bf16 per operand
    split in registers, six MFMA terms: fp32-grade; also what a plane mode falls back to on shapes the stream kernels do not
    serve), SPLIT3 (two bf16, three terms: 16 mantissa bits) and the exact fp32 MFMA."""
    from bonai_amd import kernels as K
    prev = K.F32_CONTRACT
    K.F32_CONTRACT = {'planes_f16': K.F32_PLANES_F16, 'planes_f16x4': K.F32_PLANES_F16X4, 'planes_bf16': K.F32_PLANES_BF16, 'split6': K.F32_SPLIT6,
                      'split3': K.F32_SPLIT3, 'exact': K.F32_EXACT}[request.param]
    yield request.param
    K.F32_CONTRACT = prev
