"""Weight-gradient kernel time against the split-K factor on the backbone shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K

SHAPES = [('layer3.1x1.1024-256', 8, 1024, 256, 64, 1), ('layer3.3x3', 8, 256, 256, 64, 3), ('layer2.1x1.128-512', 8, 128, 512, 128, 1),
          ('layer2.3x3', 8, 128, 128, 128, 3), ('layer4.3x3', 8, 512, 512, 32, 3)]


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, B, Cin, Cout, H, R in SHAPES:
    x = torch.randn(B, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, Cout, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * H * H * Cin * Cout * R * R / 1e9
    for var, vn in ((K.WGRAD_T128, 't128'), (K.WGRAD_RING128, 'ring128'), (K.WGRAD_STREAM256, 'stream256')):
        if var == K.WGRAD_STREAM256 and (Cin % 256 or Cout % 256):
            continue
        K.WGRAD_VARIANT = var
        row = f'{name:22s} {vn:10s}'
        for sp in (1, 2, 4, 8, 16, 32, 64, 128):
            us = timeit(lambda: K.conv2d_wgrad(g, x, R, R, 1, R // 2, splits=sp))
            row += f'  s{sp}:{us:6.1f}us/{gf / us * 1e3:5.0f}TF'
        print(row, flush=True)
K.WGRAD_VARIANT = K.WGRAD_AUTO
# split-K SLOTS (plain stores, summed by the batched unpack): the launch alone and + the slot sum's read at 3 TB/s
print('--- slots (auto kernel); launch us / + slot-sum read us')
SH2 = SHAPES + [('layer3.1x1.256-1024', 8, 256, 1024, 64, 1), ('layer2.1x1.512-128', 8, 512, 128, 128, 1), ('fpn.P3.3x3', 8, 256, 256, 128, 3),
                ('layer4.1x1.2048-512', 8, 2048, 512, 32, 1), ('layer4.1x1.512-2048', 8, 512, 2048, 32, 1)]
for name, B, Cin, Cout, H, R in SH2:
    x = torch.randn(B, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, Cout, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    row = f'{name:22s} auto-atomic {timeit(lambda: K.conv2d_wgrad(g, x, R, R, 1, R // 2)):6.1f}us |'
    for sp in (0, 2, 4, 8, 16, 32, 64):
        try:
            r = K.conv2d_wgrad(g, x, R, R, 1, R // 2, splits=sp, slots_ok=True)
            ns = r.shape[1] if r.dim() == 5 else 1
            us = timeit(lambda: K.conv2d_wgrad(g, x, R, R, 1, R // 2, splits=sp, slots_ok=True))
            row += f'  s{sp}(n{ns}):{us:6.1f}/+{ns * R * R * Cin * Cout * 4 / 3e6:4.1f}'
        except Exception as e:
            row += f'  s{sp}: {type(e).__name__}'
    print(row, flush=True)
