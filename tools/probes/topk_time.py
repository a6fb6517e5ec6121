"""Time of the RPN's two segmented top-k launches against the full sort they replaced (run on the GPU box)."""
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
    return s.elapsed_time(e) / iters * 1e3


def main():
    B = 8
    lv = [3 * 256 * 256, 3 * 128 * 128, 3 * 64 * 64, 3 * 32 * 32, 3 * 16 * 16]
    N = sum(lv)
    off = [0]
    for b in range(B):
        for n in lv:
            off.append(off[-1] + n)
    seg = torch.tensor(off, dtype=torch.int64, device='cuda')
    keys = torch.rand(B * N, device='cuda').sigmoid()
    two = timeit(lambda: K.segmented_topk_desc(keys, seg, 3000, max_segment=max(lv), seg_lengths=lv * B))
    K.TOPK_MAX_SEGMENT, keep = 1 << 30, K.TOPK_MAX_SEGMENT
    one = timeit(lambda: K.segmented_topk_desc(keys, seg, 3000, max_segment=max(lv)))
    K.TOPK_MAX_SEGMENT = keep
    print(f'anchors: top-3000 of {B} x {lv}:  two-stage {two:7.1f} us   one workgroup per segment {one:7.1f} us   full sort '
          f'{timeit(lambda: K.segmented_sort_desc(keys, seg)):7.1f} us')
    C = 12768
    seg2 = torch.arange(B + 1, dtype=torch.int64, device='cuda') * C
    k2 = torch.rand(B * C, device='cuda')
    k2[torch.rand(B * C, device='cuda') < 0.8] = -1.0
    print(f'post-NMS: top-1000 of {B} x {C}:  topk {timeit(lambda: K.segmented_topk_desc(k2, seg2, 1000, max_segment=C)):7.1f} us   full sort '
          f'{timeit(lambda: K.segmented_sort_desc(k2, seg2)):7.1f} us')


if __name__ == '__main__':
    main()
