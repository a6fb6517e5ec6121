"""Time the K-shallow 1x1 convs of the backbone (memory/latency-bound) in isolation: ms, TFLOP/s and effective HBM rate."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
SH = [(8, 64, 64, 256, 1024, True), (8, 128, 128, 128, 512, True), (8, 256, 256, 64, 256, True), (8, 32, 32, 512, 2048, True),
      (8, 64, 64, 1024, 256, False), (8, 128, 128, 512, 128, False), (8, 256, 256, 256, 64, False), (8, 256, 256, 256, 256, False)]
for B, H, W, Cin, Cout, res in SH:
    x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, 1, Cout, Cin, device='cuda').bfloat16()
    b = torch.randn(1, Cout, device='cuda')
    r = torch.randn(B, Cout, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    for _ in range(3):
        K.conv2d_fwd(x, w, b, 1, 1, 1, 0, relu=True, residual=r)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        K.conv2d_fwd(x, w, b, 1, 1, 1, 0, relu=True, residual=r)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    M = B * H * W
    byts = M * (Cin + Cout * (2 if res else 1)) * 2
    print(f'{(B,H,W,Cin,Cout,res)!s:36s} {ms*1e3:7.1f} us  {2*M*Cin*Cout/ms/1e9:7.1f} TF  {byts/ms/1e9:6.2f} TB/s')

# streaming floor for the same output/residual footprint: out = relu(a + r) as one aten launch (reads 2, writes 1 map of Cout channels)
print('--- aten add+relu floor on the Cout-channel maps (bytes = 3 maps)')
for B, H, W, Cin, Cout, res in SH[:4]:
    a = torch.randn(B, Cout, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(a)
    o = torch.empty_like(a)
    for _ in range(3):
        torch.add(a, r, out=o)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        torch.add(a, r, out=o)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f'{(B,H,W,Cout)!s:36s} {ms*1e3:7.1f} us  {3 * a.numel() * 2 / ms / 1e9:6.2f} TB/s')
print('--- same convs without residual')
for B, H, W, Cin, Cout, res in SH[:4]:
    x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, 1, Cout, Cin, device='cuda').bfloat16()
    b = torch.randn(1, Cout, device='cuda')
    for _ in range(3):
        K.conv2d_fwd(x, w, b, 1, 1, 1, 0, relu=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        K.conv2d_fwd(x, w, b, 1, 1, 1, 0, relu=True)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    M = B * H * W
    print(f'{(B,H,W,Cin,Cout)!s:36s} {ms*1e3:7.1f} us  {M * (Cin + Cout) * 2 / ms / 1e9:6.2f} TB/s')
