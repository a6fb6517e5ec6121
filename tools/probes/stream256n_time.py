"""256 x 128-cout tiles on the three-stage ring (LOFT_CONV_STREAM256N) against the shipped 256 x 256 two-stage stream tile on the
RoI-head and P2 shapes of the bench step (forward launches with bias + ReLU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K
SHAPES = [('foa.3x3', 2048, 256, 256, 7, 7, 3, 1, 1, 4), ('mask.3x3', 2048, 256, 256, 14, 14, 3, 1, 1, 1),
          ('fpn.P2.3x3', 8, 256, 256, 256, 256, 3, 1, 1, 1), ('fpn.P3.3x3', 8, 256, 256, 128, 128, 3, 1, 1, 1),
          ('fc1', 8192, 12544, 1024, 1, 1, 1, 1, 0, 1), ('layer2.1x1', 8, 128, 512, 128, 128, 1, 1, 0, 1),
          ('layer3.1x1', 8, 256, 1024, 64, 64, 1, 1, 0, 1)]
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
variants = [('auto', K.CONV_AUTO), ('stream256', K.CONV_STREAM256), ('stream256n', K.CONV_STREAM256N)]
print(f'{"shape":14s}' + ''.join(f'{n + " us":>14s}{n + " TF":>14s}' for n, _ in variants))
for name, B, Cin, Cout, H, W, R, st, pad, G in SHAPES:
    x = torch.randn(G * B if G > 1 else B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(G, Cout, Cin, R, R, device='cuda') * 0.02
    wp = torch.stack([K.pack_w_fwd(w[i]) for i in range(G)])
    bias = torch.zeros(G, Cout, device='cuda')
    gflop = 2.0 * G * B * H * W * Cout * Cin * R * R / 1e9
    best = {n: 1e9 for n, _ in variants}
    outs = {}
    for rnd in range(3):
        for n, v in variants:
            K.CONV_VARIANT = v
            try:
                best[n] = min(best[n], timeit(lambda: K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G)))
                outs[n] = K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G)
            except Exception as ex:
                best[n] = float('nan')
            finally:
                K.CONV_VARIANT = K.CONV_AUTO
    same = all(torch.equal(outs['stream256'], o) for o in outs.values()) if 'stream256' in outs else None
    print(f'{name:14s}' + ''.join(f'{best[n] * 1e3:14.1f}{gflop / best[n]:14.1f}' for n, _ in variants) + f'   bit-identical {same}', flush=True)
