"""fp32 parity mode, contraction by contraction: per-launch time of the FOA / mask / P2 shapes forward and weight gradient
(run on the GPU box):  python tools/probes/fp32_modes.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K

SH = [('foa.3x3 (4x2048 roi)', 2048, 256, 256, 3, 1, 7, 4), ('mask.3x3 (2048 roi)', 2048, 256, 256, 3, 1, 14, 1),
      ('fpn.P2.3x3', 8, 256, 256, 3, 1, 256, 1), ('layer3.3x3', 8, 256, 256, 3, 1, 64, 1), ('layer3.1x1.256-1024', 8, 256, 1024, 1, 0, 64, 1),
      ('layer2.3x3', 8, 128, 128, 3, 1, 128, 1)]
MODES = [('planes_f16', K.F32_PLANES_F16), ('planes_f16x4', K.F32_PLANES_F16X4), ('planes_bf16', K.F32_PLANES_BF16), ('split6', K.F32_SPLIT6), ('split3', K.F32_SPLIT3)]


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


print(f'{"shape":26s} {"GFLOP":>8s} ' + ' '.join(f'{n + " fwd ms":>18s} {"wgrad ms":>9s}' for n, _ in MODES))
for name, B, cin, cout, k, p, hw, G in SH:
    x = torch.randn(G * B, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(G, k * k, cout, cin, device='cuda') * 0.02
    b = torch.zeros(G, cout, device='cuda')
    g = torch.randn(G * B, cout, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-3
    gf = 2.0 * G * B * hw * hw * cout * cin * k * k / 1e9
    row = f'{name:26s} {gf:8.1f} '
    for n, code in MODES:
        K.F32_CONTRACT = code
        t1 = timeit(lambda: K.conv2d_fwd(x, w, b, k, k, 1, p, relu=True, out_dtype=torch.float32, groups=G))
        t2 = timeit(lambda: K.conv2d_wgrad(g, x, k, k, 1, p, groups=G))
        row += f'{t1:9.3f} ({gf / t1:6.0f}TF) {t2:9.3f} '
    print(row, flush=True)
# the split passes alone
for n, dt in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    x = torch.randn(4 * 2048, 256, 7, 7, device='cuda').contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: K.split_planes(x, dt))
    print(f'split_planes {n} of {x.numel() * 4 / 1e6:.0f} MB: {t:.3f} ms')
