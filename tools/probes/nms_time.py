"""Time of the RPN's segmented NMS launch pair (mask + scan) at the training step's geometry: 8 images x 5 levels, up to 3000 boxes
per segment, boxes spread like a random-init RPN's proposals (few suppressions: the scan's worst case) and clustered like a
trained one's.   usage: python nms_time.py"""
import os, sys
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


def boxes(n, spread, g):
    c = torch.rand(n, 2, generator=g) * 1024 if spread else (torch.randint(0, 80, (n,), generator=g)[:, None] * 12.5 + torch.randn(n, 2, generator=g) * 4)
    wh = torch.rand(n, 2, generator=g) * 60 + 8
    return torch.cat([c - wh / 2, c + wh / 2], 1)


def main():
    g = torch.Generator().manual_seed(0)
    lens = [3000, 3000, 3000, 3000, 768] * 8
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int64, device='cuda')
    for name, spread in (('spread (random-init RPN)', True), ('clustered (trained RPN)', False)):
        b = boxes(sum(lens), spread, g).cuda()
        keep = K.nms_segmented(b, off, 0.7, max_segment=3000)
        t = timeit(lambda: K.nms_segmented(b, off, 0.7, max_segment=3000))
        print(f'{name:28s} mask + scan {t:7.1f} us   kept {int(keep.sum())} of {sum(lens)}')


if __name__ == '__main__':
    main()
