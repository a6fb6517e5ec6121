"""Why is the back-to-back step slower than a step followed by a device sync?  Variants of the timing loop + allocator stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(8):
    tr.train_step(data)
torch.cuda.synchronize()
def loop(n, every):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(n):
        tr.train_step(data)
        if every and (it + 1) % every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    st1 = torch.cuda.memory_stats()
    return dt, st1['num_device_alloc'] - st0['num_device_alloc'], st1['num_alloc_retries'] - st0['num_alloc_retries'], \
        st1['reserved_bytes.all.peak'] / 2**30
for rnd in range(2):
    for every in (0, 1, 2, 4):
        dt, nalloc, nretry, peak = loop(12, every)
        print(f'sync every {every or "never":>5}: {dt:6.2f} ms/step   hipMalloc calls {nalloc}  retries {nretry}  reserved peak {peak:.1f} GiB', flush=True)
