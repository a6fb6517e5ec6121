"""How much of an RoI-map stream-conv launch is lost to the last, partly filled round of workgroups?
TFLOP/s of the FOA (7x7, 4 groups) and mask-head (14x14) 3x3 launches over RoI counts that give 5.4 .. 7.0 rounds of
256-row tiles on 256 CUs (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K


def timeit(fn, iters=20):
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
    variant = int(os.environ.get('CONV_VARIANT', '0'))
    print(f'{"shape":10s} {"RoIs":>6s} {"tiles":>6s} {"rounds":>7s} {"us":>8s} {"TF alg":>8s} {"us/round":>9s}')
    for name, P, G in (('foa', 7, 4), ('mask', 14, 1)):
        for B in (256, 1024, 1536, 1792, 2048, 2304, 2560, 3072):
            if P == 14:
                Bm = B // 4 * 1  # mask maps: 4x the positions, so a quarter of the RoIs gives the same tile count
                Bm = B
            x = torch.randn(G * B, 256, P, P, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
            w = torch.randn(G, 256, 256, 3, 3, device='cuda') * 0.02
            wp = torch.stack([K.pack_w_fwd(w[i]) for i in range(G)])
            bias = torch.zeros(G, 256, device='cuda')
            K.CONV_VARIANT = variant
            ms = min(timeit(lambda: K.conv2d_fwd(x, wp, bias, 3, 3, 1, 1, relu=True, groups=G)) for _ in range(3))
            K.CONV_VARIANT = K.CONV_AUTO
            nb = max(1, B // 256)
            S = (B + nb - 1) // nb
            tiles = G * ((nb * P * P * S + 255) // 256)
            gflop = 2.0 * G * B * P * P * 256 * 256 * 9 / 1e9
            line = f'{name:10s} {B:6d} {tiles:6d} {tiles / 256:7.3f} {ms * 1e3:8.1f} {gflop / ms:8.1f} {ms * 1e3 / (tiles / 256):9.2f}'
            if os.environ.get('WGRAD'):          # the weight-gradient launch of the same layer (RoI-map instance of the stream kernel)
                g = torch.randn_like(x)
                wms = min(timeit(lambda: K.conv2d_wgrad(g, x, 3, 3, 1, 1, groups=G)) for _ in range(3))
                line += f'   wgrad {wms * 1e3:8.1f} us {gflop / wms:8.1f} TF'
            print(line, flush=True)


if __name__ == '__main__':
    main()
