"""Which python frames issue aten::copy_/clone/contiguous during one train step (finds stray layout copies)."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(model)
data = make_batch(4, 512, 20, device='cuda')
for _ in range(2):
    tr.train_step(data)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_step(data)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::add', 'aten::add_', 'aten::zeros_like', 'aten::zero_', 'aten::fill_'):
        shp = ev.input_shapes[0] if ev.input_shapes else None
        if shp and len(shp) == 4 and shp[0] * shp[1] * shp[2] * shp[3] >= 1 << 18:
            cnt[(ev.name, tuple(shp))] += 1
for (name, st), n in sorted(cnt.items()):
    print(n, name, st)
