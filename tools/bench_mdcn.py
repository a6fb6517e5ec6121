"""Micro-benchmark of the DCNv2 sampling kernels at the LOFT config-4 shapes (HIP-event timing, HBM roofline).
python tools/bench_mdcn.py [--zero-offsets]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bonai_amd import kernels as K  # noqa: E402

SHAPES = [('fpn P2 3x3', 8, 256, 256, 3, 1), ('fpn P3 3x3', 8, 256, 128, 3, 1), ('layer2 3x3', 8, 128, 128, 3, 1),
          ('layer2.0 3x3 s2', 8, 128, 256, 3, 2), ('layer3 3x3', 8, 256, 64, 3, 1), ('layer4 3x3', 8, 512, 32, 3, 1),
          ('lateral C3 1x1', 8, 512, 128, 1, 1), ('lateral C5 1x1', 8, 2048, 32, 1, 1)]


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--zero-offsets', action='store_true')
    args = ap.parse_args()
    torch.manual_seed(0)
    for name, B, C, H, k, s in SHAPES:
        pad = k // 2
        OH = (H + 2 * pad - k) // s + 1
        KK = k * k
        x = torch.randn(B, C, H, H, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        omc = (3 * KK + 3) // 4 * 4
        om = torch.randn(B, omc, OH, OH, device='cuda').contiguous(memory_format=torch.channels_last)
        if args.zero_offsets:
            om.zero_()
        col = K.mdcn_sample_fwd(x, om, k, k, s, pad)
        dcol = torch.randn_like(col)
        tf = timeit(lambda: K.mdcn_sample_fwd(x, om, k, k, s, pad))
        tb = timeit(lambda: K.mdcn_sample_bwd(x, om, dcol, k, k, s, pad))
        M = B * OH * OH
        bytes_f = M * KK * C * 2 + x.numel() * 2 + M * 3 * KK * 4                  # col written + x read once + offsets
        bytes_b = M * KK * C * 2 + x.numel() * 2 + x.numel() * 4 + 2 * M * 3 * KK * 4   # dcol + x read, dx written, offsets r/w
        print(f'{name:18s} M={M:7d} C={C:4d}  fwd {tf * 1e3:8.1f} us ({bytes_f / tf / 1e6:7.1f} GB/s)   '
              f'bwd {tb * 1e3:8.1f} us ({bytes_b / tb / 1e6:7.1f} GB/s)')


if __name__ == '__main__':
    main()
