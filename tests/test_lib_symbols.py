"""CPU: the C-ABI library loads and exports every symbol include/loft_hip.h declares."""
import ctypes
import os

import torch  # noqa: F401  (load torch's HIP runtime before libloft_hip.so)

from bonai_amd import lib as L


def test_header_symbols_exported():
    names = L.exported_symbols()
    assert len(names) >= 5
    if not (os.path.exists(L._LIB_PATH) and os.path.exists(L._LIB_PATH_F16)):
        from bonai_amd import build
        build.build()
    for path, code in ((L._LIB_PATH, L.BF16), (L._LIB_PATH_F16, L.F16)):      # the bfloat16 and the binary16 build: same C-ABI
        cdll = ctypes.CDLL(path)
        missing = [n for n in names if not hasattr(cdll, n)]
        assert not missing, f'symbols declared in include/loft_hip.h but not exported by {path}: {missing}'
        assert cdll.loft_act16_dtype() == code


def test_act16_mode_switch():
    import pytest
    assert L.act16() == torch.bfloat16
    prev = L.set_act16(torch.float16)
    try:
        assert prev == torch.bfloat16 and L.act16() == torch.float16
        assert L.load().loft_act16_dtype() == L.F16
    finally:
        L.set_act16(prev)
    assert L.load().loft_act16_dtype() == L.BF16
    with pytest.raises(L.LoftHipError):
        L.set_act16(torch.float32)


def test_no_cpu_fallback():
    import pytest
    import torch
    from bonai_amd import kernels as K
    rois = torch.zeros(1, 5)
    feat = torch.zeros(1, 4, 8, 8).contiguous(memory_format=torch.channels_last)
    with pytest.raises(L.LoftHipError):
        K.roi_align_fwd([feat], rois, 7, [4])


def test_library_sources_read_no_environment():
    """The kernel library takes its variants as explicit arguments (the *_v entry points of include/loft_hip.h): no getenv in csrc."""
    import glob
    csrc = os.path.join(os.path.dirname(os.path.abspath(L.__file__)), 'csrc')
    offenders = [f for f in glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')) if 'getenv' in open(f).read()]
    assert not offenders, offenders


def test_selector_constants_match_the_header():
    """bonai_amd.kernels mirrors the kernel selectors of include/loft_hip.h (LOFT_CONV_* / LOFT_WGRAD_* / LOFT_ROI_* / LOFT_F32_*
    and the LOFT_CONV_FLAG_* bits) as plain integers: every mirrored name must carry the header's value."""
    import re
    from bonai_amd import kernels as K
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'loft_hip.h')).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r'^#define\s+(LOFT_[A-Z0-9_x]+)\s+(0x[0-9a-fA-F]+|\d+)\b', hdr, re.M)}
    checked = 0
    for name, value in defs.items():
        for prefix in ('LOFT_CONV_', 'LOFT_WGRAD_', 'LOFT_ROI_', 'LOFT_F32_'):
            if name.startswith(prefix):
                py = name[len('LOFT_'):]
                if hasattr(K, py):
                    assert getattr(K, py) == value, (name, value, getattr(K, py))
                    checked += 1
    assert checked >= 30, checked
    for must in ('F32_SPLIT6', 'F32_SPLIT3', 'F32_EXACT', 'ROI_FWD_SEP4', 'ROI_BWD_PIPE', 'CONV_STREAM256N'):
        assert 'LOFT_' + must in defs and getattr(K, must) == defs['LOFT_' + must], must
