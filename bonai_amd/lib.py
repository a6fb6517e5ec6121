"""ctypes binding of libloft_hip.so (include/loft_hip.h) -- the ONLY compute backend.

There is deliberately no CPU or eager-PyTorch fallback: if the shared library is missing, or an
op is called with tensors that are not on a HIP device, this module raises.
"""
import ctypes
import os

import torch

# LOFT_HIP_LIB: an alternative build of the same library (A/B timing of kernel variants on one box)
_LIB_PATH = os.environ.get('LOFT_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libloft_hip.so')
_lib = None

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class LoftHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the CDLL.  Raises LoftHipError when the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise LoftHipError(
                f'{_LIB_PATH} is missing: build it with `python -m bonai_amd.build` '
                '(hipcc --offload-arch=gfx950).  bonai_amd has no CPU / eager fallback.')
        lib = ctypes.CDLL(_LIB_PATH)
        lib.loft_nms_workspace_bytes.restype = c_int64
        lib.loft_nms_workspace_bytes.argtypes = [c_int64, c_int64]
        lib.loft_soft_nms_workspace_bytes.restype = c_int64
        lib.loft_random_sample_workspace_bytes.restype = c_int64
        lib.loft_conv_wgrad_patch_workspace_bytes.restype = c_int64
        lib.loft_soft_nms_workspace_bytes.argtypes = [c_int64]
        lib.loft_mdcn_bwd_workspace_bytes.restype = c_int64
        _lib = lib
    return _lib


def exported_symbols():
    """Names declared in include/loft_hip.h (parsed), for the symbol-presence test."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'loft_hip.h')
    text = open(hdr).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(loft_[a-z0-9_]+)\s*\(', text)))


def check(code, what):
    if code != 0:
        raise LoftHipError(f'{what} failed with hipError_t {code}')


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


_DEV_INDEX = None


def stream():
    """hipStream_t of torch's CURRENT stream on this process's device (one device per process: one process per GPU).
    torch.cuda.current_stream() builds a Stream object through several python layers (~10 us; 900-3000 calls per step);
    the raw-pointer query is the same information in well under a microsecond and still follows torch.cuda.stream(...)."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return c_void_p(torch._C._cuda_getCurrentRawStream(_DEV_INDEX))


def dev_check(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise LoftHipError('bonai_amd ops need HIP device tensors (no CPU fallback); got a CPU tensor')


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise LoftHipError(f'unsupported dtype {t.dtype}')


def arr(ctype, values):
    return (ctype * len(values))(*values)
