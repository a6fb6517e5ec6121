"""One shape of the 256 x 256 stream kernel, a few launches, for rocprofv3 --pmc passes (tools/pmc_stream.py).
    python tools/stream_shape.py [foa|mask|p2] [variant code] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonai_amd import kernels as K

SHAPES = {'p2': (8, 256, 256, 256, 256, 3, 1, 1, 1), 'mask': (872, 256, 256, 14, 14, 3, 1, 1, 1), 'foa': (3488, 256, 256, 7, 7, 3, 1, 1, 4)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'foa'
    var = int(sys.argv[2], 0) if len(sys.argv) > 2 else K.CONV_STREAM256
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    B, Cin, Cout, H, W, R, st, pad, G = SHAPES[name]
    x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(G, Cout, Cin, R, R, device='cuda') * 0.02
    wp = torch.stack([K.pack_w_fwd(w[i]) for i in range(G)])
    bias = torch.zeros(G, Cout, device='cuda')
    K.CONV_VARIANT = var
    for _ in range(iters):
        K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
