"""CPU: the C-ABI library loads and exports every symbol include/loft_hip.h declares."""
import ctypes
import os

import torch  # noqa: F401  (load torch's HIP runtime before libloft_hip.so)

from bonai_amd import lib as L


def test_header_symbols_exported():
    names = L.exported_symbols()
    assert len(names) >= 5
    if not os.path.exists(L._LIB_PATH):
        from bonai_amd import build
        build.build()
    cdll = ctypes.CDLL(L._LIB_PATH)
    missing = [n for n in names if not hasattr(cdll, n)]
    assert not missing, f'symbols declared in include/loft_hip.h but not exported: {missing}'


def test_no_cpu_fallback():
    import pytest
    import torch
    from bonai_amd import kernels as K
    rois = torch.zeros(1, 5)
    feat = torch.zeros(1, 4, 8, 8).contiguous(memory_format=torch.channels_last)
    with pytest.raises(L.LoftHipError):
        K.roi_align_fwd([feat], rois, 7, [4])
