"""Time the 64 -> 64 channel 3x3 conv of HRNet's high-resolution branch in isolation."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
for (B, H, W, res) in [(8, 256, 256, False), (8, 256, 256, True), (8, 128, 128, True)]:
    x = torch.randn(B, 64, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, 9, 64, 64, device='cuda').bfloat16()
    b = torch.randn(1, 64, device='cuda')
    r = torch.randn_like(x) if res else None
    for _ in range(3):
        K.conv2d_fwd(x, w, b, 3, 3, 1, 1, relu=True, residual=r)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        K.conv2d_fwd(x, w, b, 3, 3, 1, 1, relu=True, residual=r)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    byts = B * H * W * 64 * 2 * (3 if res else 2)
    print(f'{(B,H,W,res)!s:24s} {ms*1e3:7.1f} us  {byts/ms/1e9:6.2f} TB/s')
