import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K, nn as F2
from bonai_amd.debug import DBG
x = torch.randn(2048, 256, 14, 14, device='cuda').to(K.L.act16()).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 2, 2, device='cuda') * 0.05
b = torch.zeros(256, device='cuda')
pre = F2.narrow_head_prepack(torch.randn(1, 256, device='cuda') * 0.1, torch.randn(1, device='cuda'), x.dtype)
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rnd in range(2):
    for off in (False, True):
        with DBG.override(no_deconv_fusion=off), torch.no_grad():
            print('four launches' if off else 'one launch   ', 'plain %.1f us' % timeit(lambda: F2.deconv2x2_relu(x, w, b)), 'with head %.1f us' % timeit(lambda: F2.deconv2x2_relu(x, w, b, head=pre)))
