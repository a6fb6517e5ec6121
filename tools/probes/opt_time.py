"""Time of the optimizer's two launches on the LOFT R50 arena size (41.4 M fp32 parameters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K
n = 41_400_000
p, g, m = (torch.randn(n, device='cuda') for _ in range(3))
out = torch.zeros(1, device='cuda')


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


t1 = timeit(lambda: K.sumsq_(g, out))
t2 = timeit(lambda: K.sgd_momentum_(p, g, m, out, 35.0, 1e-3, 0.9, 1e-4))
print(f'sumsq {t1:.1f} us ({n * 4 / t1 / 1e6:.2f} TB/s)   sgd {t2:.1f} us ({n * 20 / t2 / 1e6:.2f} TB/s)')
out.zero_(); K.sumsq_(g, out)
print('sumsq rel err', abs(float(out) - float((g.double() ** 2).sum())) / float((g.double() ** 2).sum()))
