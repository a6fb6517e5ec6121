"""Per-workgroup time line of the RoIAlign forward (trace build of the library: tools/probes/_abl/libtrace.so, -DRF8_TRACE=1, where
the `order` argument is a [K][8] u64 stamp buffer).  LOFT_HIP_LIB must point at the trace build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bonai_amd import kernels as K, lib as L
from bonai_amd.synth import make_batch
assert 'trace' in os.environ.get('LOFT_HIP_LIB', '')
data = make_batch(8, 1024, 80, device='cuda')
g = torch.Generator().manual_seed(7)
rois = []
for i, gb in enumerate(data['gt_bboxes']):
    b = gb.cpu()
    wh = b[:, 2:] - b[:, :2]
    reps = [(b + (torch.rand(b.shape[0], 4, generator=g) - 0.5) * 0.16 * torch.cat([wh, wh], 1)).clamp(0, 1024) for _ in range(4)]
    jb = torch.cat(reps, 0)[:256]
    rois.append(torch.cat([torch.full((jb.shape[0], 1), float(i)), jb], 1))
rois = torch.cat(rois).cuda()
strides = [4, 8, 16, 32]
feats = [torch.randn(8, 1024 // s, 1024 // s, 256, device='cuda').bfloat16().permute(0, 3, 1, 2) for s in strides]
lib = L.load()
for P, n_rot, var in ((14, 1, 0), (7, 4, 0), (7, 1, 0), (14, 1, 255 << 8)):
    Kn = rois.shape[0]
    out = K.empty_nhwc(n_rot * Kn, 256, P, P, torch.bfloat16, rois.device)
    H, W, S = K._level_args(feats, strides)
    fp = L.arr(K.c_void_p, [f.data_ptr() for f in feats])
    tr = torch.zeros(Kn, 8, dtype=torch.int64, device='cuda')
    for _ in range(3):
        L.check(lib.loft_roi_align_fwd_ord(fp, H, W, S, 4, 56, 256, L.dtype_code(feats[0]), L.ptr(rois), Kn, P, n_rot, L.ptr(out), var,
                                           L.ptr(tr), L.stream()), 'fwd')
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    L.check(lib.loft_roi_align_fwd_ord(fp, H, W, S, 4, 56, 256, L.dtype_code(feats[0]), L.ptr(rois), Kn, P, n_rot, L.ptr(out), var,
                                       L.ptr(tr), L.stream()), 'fwd')
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    t = tr.cpu().numpy().astype(np.int64)
    hw = t[:, 6]
    xcc = (hw >> 32) & 0xf
    # every XCD has its own s_memtime base: normalise per XCD, calibrate the tick on the kernel's duration
    spans = []
    for xc in np.unique(xcc):
        m = xcc == xc
        b = t[m, 0].min()
        t[m, 0:5] -= b
        spans.append(t[m, 4].max())
    t0 = 0
    span = float(np.median(spans))
    tick = us / span
    path = t[:, 5] & 0xffff
    Fh, Fw = (t[:, 5] >> 16) & 0xff, (t[:, 5] >> 24) & 0xff
    print(f'P={P} n_rot={n_rot} variant={var:#x} K={Kn}: {us:.1f} us, span {span} ticks -> {tick * 1e3:.2f} ns / tick')
    for name, sel in (('sample', path == 0), ('stream', path == 1), ('lds', path >= 64)):
        if not sel.any():
            continue
        d = t[sel]
        seg = lambda a, b: (d[:, b] - d[:, a]) * tick
        print(f'  {name:7s} n={sel.sum():5d} footprint {Fh[sel].mean():.1f} x {Fw[sel].mean():.1f}  geom {seg(0, 1).mean():.2f} us, tables '
              f'{(seg(1, 2).mean() if name != "sample" else 0):.2f}, stage {(seg(2, 3).mean() if name == "lds" else 0):.2f}, '
              f'main {(seg(3, 4).mean() if name == "lds" else seg(2, 4).mean() if name == "stream" else seg(1, 4).mean()):.2f}, '
              f'whole {seg(0, 4).mean():.2f} (p90 {np.percentile(seg(0, 4), 90):.2f}) us')
    starts = np.sort((t[:, 0] - t0) * tick)
    print('  workgroup start times (us): ' + ' '.join(f'{starts[int(q * (Kn - 1))]:.0f}' for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)))
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1)
    # concurrency: mean number of workgroups alive per CU id while the kernel runs
    alive = ((t[:, 4] - t[:, 0]) * tick).sum() / us / len(np.unique(cu))
    print(f'  distinct (xcc, se, cu) ids {len(np.unique(cu))}, mean workgroups alive per id {alive:.2f}')
