"""The three blocks of the frozen layer1 at the bench size (8 x 256 x 256 maps): fused tail (loft_bneck_tail_bf16) against the
unfused tap-conv launches (DBG.no_bneck_fusion)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd.debug import DBG
from bonai_amd.loft.backbone import Bottleneck
torch.manual_seed(0)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot = [0.0, 0.0]
for name, cin, ds in (('layer1.0 (conv shortcut)', 64, True), ('layer1.1', 256, False), ('layer1.2', 256, False)):
    blk = Bottleneck(cin, 64, stride=1, downsample=ds).cuda()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 4:
                p.copy_(torch.randn_like(p) * 0.05)
    blk.requires_grad_(False)
    x = torch.relu(torch.randn(8, cin, 256, 256, device='cuda')).bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        f = timeit(lambda: blk(x))
        with DBG.override(no_bneck_fusion=True):
            u = timeit(lambda: blk(x))
    tot[0] += f; tot[1] += u
    print(f'{name:26s} fused {f:7.1f} us   unfused {u:7.1f} us')
print(f'{"layer1":26s} fused {tot[0]:7.1f} us   unfused {tot[1]:7.1f} us')
