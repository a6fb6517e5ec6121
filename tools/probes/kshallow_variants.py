"""K-shallow 1x1 convs of the backbone: every applicable kernel variant, forward (+ residual + ReLU) and without residual;
us per launch and effective HBM rate of the algorithmic bytes (run on the GPU box)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
SH = [(8, 64, 64, 256, 1024), (8, 128, 128, 128, 512), (8, 256, 256, 64, 256), (8, 32, 32, 512, 2048),
      (8, 64, 64, 1024, 256), (8, 128, 128, 512, 128), (8, 256, 256, 256, 64), (8, 32, 32, 2048, 512)]
VAR = [('auto', K.CONV_AUTO), ('stream', K.CONV_STREAM256), ('t256f', K.CONV_T256_FAST), ('t128s', K.CONV_T128_SINGLE),
       ('t128f', K.CONV_T128_FAST), ('t128', K.CONV_T128), ('t128x64', K.CONV_T128x64), ('s128', K.CONV_STREAM128), ('s64', K.CONV_STREAM64),
       ('s64n', K.CONV_STREAM64N), ('s256n', K.CONV_STREAM256N), ('ring32', K.CONV_RING32)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for res in (True, False):
    print('--- with residual + relu' if res else '--- relu only')
    print(f'{"shape":30s}' + ''.join(f'{n:>10s}' for n, _ in VAR) + '   floor(5.5TB/s)')
    for B, H, W, Cin, Cout in SH:
        x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(1, 1, Cout, Cin, device='cuda').bfloat16()
        b = torch.randn(1, Cout, device='cuda')
        r = torch.randn(B, Cout, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last) if res else None
        row = f'{(B, H, W, Cin, Cout)!s:30s}'
        for n, v in VAR:
            K.CONV_VARIANT = v
            try:
                us = timeit(lambda: K.conv2d_fwd(x, w, b, 1, 1, 1, 0, relu=True, residual=r))
                row += f'{us:10.1f}'
            except Exception:
                row += f'{"-":>10s}'
            K.CONV_VARIANT = K.CONV_AUTO
        M = B * H * W
        byts = M * (Cin + Cout * (2 if res else 1)) * 2
        print(row + f'   {byts / 5.5e12 * 1e6:6.1f}', flush=True)
