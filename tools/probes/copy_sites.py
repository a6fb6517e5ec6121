"""Where do the step's device-to-device copies, fills and stock elementwise kernels come from?  torch.profiler with Python
stacks over two train steps; prints per (kernel family, innermost repo frame): launches per step and device microseconds."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(4):
    tr.train_step(data)
torch.cuda.synchronize()
NSTEP = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(NSTEP):
        tr.train_step(data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith('aten::'):
        continue
    if ev.cpu_children and any(c.device_time_total > 0 and c.name.startswith('aten::') for c in ev.cpu_children):
        continue                                   # count the leaf op only
    site = str([tuple(x) if isinstance(x, (list, tuple)) else x for x in (ev.input_shapes or [])][:3])
    for fr in ev.stack or []:
        if '/bonai_amd/' in fr or '/bench.py' in fr:
            site = fr.split('/bonai_amd/')[-1] if '/bonai_amd/' in fr else fr
            break
    agg[(ev.name, site)][0] += 1
    agg[(ev.name, site)][1] += ev.device_time_total
tot = collections.defaultdict(lambda: [0, 0.0])
for (name, site), (n, us) in agg.items():
    tot[name][0] += n; tot[name][1] += us
print('--- per aten op (per step)')
for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{name:34s} n={n / NSTEP:6.1f}  {us / NSTEP:8.1f} us')
print('--- per (op, call site) (per step)')
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f'{name:26s} n={n / NSTEP:5.1f} {us / NSTEP:8.1f} us  {site[:110]}')
