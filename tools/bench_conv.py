"""Micro-benchmark of the MFMA tap-conv kernels on LOFT's layer shapes (run on the GPU box)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonai_amd import kernels as K

SHAPES = [
    # name, B, Cin, Cout, H, W, R, stride, pad, groups
    ('layer1.3x3', 8, 64, 64, 256, 256, 3, 1, 1, 1),
    ('layer1.1x1.64-256', 8, 64, 256, 256, 256, 1, 1, 0, 1),
    ('layer1.1x1.256-64', 8, 256, 64, 256, 256, 1, 1, 0, 1),
    ('layer2.3x3', 8, 128, 128, 128, 128, 3, 1, 1, 1),
    ('layer2.1x1.512-128', 8, 512, 128, 128, 128, 1, 1, 0, 1),
    ('layer3.3x3', 8, 256, 256, 64, 64, 3, 1, 1, 1),
    ('layer3.1x1.1024-256', 8, 1024, 256, 64, 64, 1, 1, 0, 1),
    ('layer3.1x1.256-1024', 8, 256, 1024, 64, 64, 1, 1, 0, 1),
    ('layer2.1x1.128-512', 8, 128, 512, 128, 128, 1, 1, 0, 1),
    ('fpn.lat.512-256', 8, 512, 256, 128, 128, 1, 1, 0, 1),
    ('fpn.lat.256-256', 8, 256, 256, 256, 256, 1, 1, 0, 1),
    ('layer4.3x3', 8, 512, 512, 32, 32, 3, 1, 1, 1),
    ('layer4.1x1.512-2048', 8, 512, 2048, 32, 32, 1, 1, 0, 1),
    ('fpn.P5.3x3', 8, 256, 256, 32, 32, 3, 1, 1, 1),
    ('fpn.P2.3x3', 8, 256, 256, 256, 256, 3, 1, 1, 1),
    ('fpn.P3.3x3', 8, 256, 256, 128, 128, 3, 1, 1, 1),
    ('mask.3x3(872roi)', 872, 256, 256, 14, 14, 3, 1, 1, 1),
    ('foa.3x3(4x872roi)', 3488, 256, 256, 7, 7, 3, 1, 1, 4),
    ('rpn.P2.3x3+fpn', 8, 256, 256, 256, 256, 3, 1, 1, 1),
    ('fc1(8192x12544x1024)', 8192, 12544, 1024, 1, 1, 1, 1, 0, 1),
]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    which = sys.argv[1:] or ['fwd', 'dgrad', 'wgrad', 'torch']
    print(f'{"shape":28s} {"GFLOP":>8s} ' + ' '.join(f'{w + " ms":>10s} {w + " TF":>8s}' for w in which))
    for name, B, Cin, Cout, H, W, R, st, pad, G in SHAPES:
        x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(G, Cout, Cin, R, R, device='cuda') * 0.02
        wp = torch.stack([K.pack_w_fwd(w[i]) for i in range(G)])
        wpt = torch.stack([K.pack_w_dgrad(w[i]) for i in range(G)])
        bias = torch.zeros(G, Cout, device='cuda')
        y = K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G)
        OH, OW = y.shape[2:]
        gflop = 2.0 * B * OH * OW * Cout * Cin * R * R / 1e9
        g = torch.randn_like(y)
        row = f'{name:28s} {gflop:8.1f} '
        for wh in which:
            if wh in ('pipe', 'stream', 'stream128', 'stream64', 'stream64n'):
                if Cout % 256:
                    row += f'{"-":>10s} {"-":>8s} '
                    continue
                K.CONV_VARIANT = K.CONV_PIPE256 if wh == 'pipe' else (K.CONV_STREAM128 if wh == 'stream128' else (K.CONV_STREAM64 if wh == 'stream64' else (K.CONV_STREAM64N if wh == 'stream64n' else K.CONV_STREAM256)))
                ms = timeit(lambda: K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G))
                K.CONV_VARIANT = K.CONV_AUTO
            elif wh in ('pipe_dgrad', 'stream_dgrad'):
                if Cin % 256 or st != 1:
                    row += f'{"-":>10s} {"-":>8s} '
                    continue
                K.CONV_VARIANT = K.CONV_PIPE256 if wh == 'pipe_dgrad' else K.CONV_STREAM256
                ms = timeit(lambda: K.conv2d_dgrad(g, wpt, (H, W), R, R, st, pad, groups=G))
                K.CONV_VARIANT = K.CONV_AUTO
            elif wh == 'fwd':
                ms = timeit(lambda: K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G))
            elif wh == 'dgrad':
                ms = timeit(lambda: K.conv2d_dgrad(g, wpt, (H, W), R, R, st, pad, groups=G))
            elif wh in ('wgrad_stream', 'wgrad_t256'):
                if Cin % 256 or Cout % 256:
                    row += f'{"-":>10s} {"-":>8s} '
                    continue
                K.WGRAD_VARIANT = K.WGRAD_STREAM256 if wh == 'wgrad_stream' else K.WGRAD_T256
                ms = timeit(lambda: K.conv2d_wgrad(g, x, R, R, st, pad, groups=G))
                K.WGRAD_VARIANT = K.WGRAD_AUTO
            elif wh == 'wgrad':
                if Cin % 128 or Cout % 128:
                    row += f'{"-":>10s} {"-":>8s} '
                    continue
                dw = torch.zeros(G, R * R, Cout, Cin, device='cuda')
                ms = timeit(lambda: K.conv2d_wgrad(g, x, R, R, st, pad, groups=G))
            else:
                if G > 1:
                    row += f'{"-":>10s} {"-":>8s} '
                    continue
                w0 = w[0].bfloat16().contiguous(memory_format=torch.channels_last)
                ms = timeit(lambda: torch.nn.functional.conv2d(x, w0, None, st, pad))
            row += f'{ms:10.3f} {gflop / ms:8.1f} '
        print(row, flush=True)


if __name__ == '__main__':
    main()
