"""conv64_patch_kernel at config 5's sizes (64 -> 64 channels, 3x3, 8 x 256^2 and 8 x 128^2; bias + ReLU, and with a residual):
microseconds per launch against the HBM floor (operands once at 6 TB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K


def timeit(fn, iters=30):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B, H in ((8, 256), (8, 128), (8, 250)):
    x = torch.randn(B, 64, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, device='cuda') * 0.05
    wp = K.pack_w_fwd(w)[None]
    bias = torch.randn(1, 64, device='cuda')
    res = torch.randn_like(x)
    byt = x.numel() * 2
    t0 = timeit(lambda: K.conv2d_fwd(x, wp, bias, 3, 3, 1, 1, relu=True))
    t1 = timeit(lambda: K.conv2d_fwd(x, wp, bias, 3, 3, 1, 1, relu=True, residual=res))
    gf = 2.0 * B * H * H * 64 * 64 * 9 / 1e9
    print(f'{B} x {H}^2: bias+relu {t0:7.1f} us ({gf / t0 * 1e-3:6.1f} TFLOP/s, floor {2 * byt / 6e6:5.1f} us)   + residual {t1:7.1f} us (floor {3 * byt / 6e6:5.1f} us)')
