"""ctypes binding of libloft_hip.so / libloft_hip_f16.so (include/loft_hip.h) -- the ONLY compute backend.

The library exists in two builds of the same sources (bonai_amd/build.py): 16-bit type bfloat16 (default) or IEEE binary16.
``set_act16(torch.float16)`` switches the process to the second one (the reference's fp16 configs); ``load()`` returns the
library of the current mode and every 16-bit tensor handed to a kernel must be of that type (the library rejects the other).

There is deliberately no CPU or eager-PyTorch fallback: if the shared library is missing, or an
op is called with tensors that are not on a HIP device, this module raises.
"""
import ctypes
import os

import torch

# LOFT_HIP_LIB: an alternative build of the same library (A/B timing of kernel variants on one box)
_LIB_PATH = os.environ.get('LOFT_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libloft_hip.so')
_LIB_PATH_F16 = os.environ.get('LOFT_HIP_LIB_F16') or os.path.join(os.path.dirname(_LIB_PATH), 'libloft_hip_f16.so')
_lib = None
_libs = {}
_act16 = torch.bfloat16

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class LoftHipError(RuntimeError):
    pass


def act16():
    """The 16-bit activation / operand dtype of the current mode (torch.bfloat16 | torch.float16)."""
    return _act16


def set_act16(dtype):
    """Switch the process between the bfloat16 and the binary16 build of the library.  Returns the previous dtype."""
    global _act16, _lib
    if dtype not in (torch.bfloat16, torch.float16):
        raise LoftHipError(f'the 16-bit type is torch.bfloat16 or torch.float16, not {dtype}')
    prev, _act16 = _act16, dtype
    _lib = None
    return prev


def load_for(dtype):
    """The CDLL of the build whose 16-bit type is `dtype`, whatever the process's current mode (the fp32 parity mode's operand
    planes are binary16 in a bfloat16 process: both libraries are then mapped).  Raises LoftHipError when it is not built."""
    path = _LIB_PATH if dtype == torch.bfloat16 else _LIB_PATH_F16
    lib = _libs.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise LoftHipError(
                f'{path} is missing: build it with `python -m bonai_amd.build` '
                '(hipcc --offload-arch=gfx950).  bonai_amd has no CPU / eager fallback.')
        lib = ctypes.CDLL(path)
        lib.loft_nms_workspace_bytes.restype = c_int64
        lib.loft_nms_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
        lib.loft_soft_nms_workspace_bytes.restype = c_int64
        lib.loft_random_sample_workspace_bytes.restype = c_int64
        lib.loft_conv_wgrad_patch_workspace_bytes.restype = c_int64
        lib.loft_soft_nms_workspace_bytes.argtypes = [c_int64]
        lib.loft_mdcn_bwd_workspace_bytes.restype = c_int64
        if lib.loft_act16_dtype() != _DT[dtype]:
            raise LoftHipError(f'{path} was built for another 16-bit type (loft_act16_dtype() = {lib.loft_act16_dtype()})')
        _libs[path] = lib
    return lib


def load():
    """Load (once per build) and return the CDLL of the current 16-bit mode.  Raises LoftHipError when it is not built."""
    global _lib
    if _lib is None:
        _lib = load_for(_act16)
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
