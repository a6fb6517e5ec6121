"""Dev helper: a few full LOFT training steps on synthetic tiles, prints losses and step time."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--gt', type=int, default=10)
ap.add_argument('--steps', type=int, default=3)
args = ap.parse_args()
cfg = Config.fromfile(os.path.join(os.path.dirname(__file__), '..', 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda()
model.train()
tr = Trainer(model, lr=0.005)
data = make_batch(args.batch, args.size, args.gt, device='cuda')
for s in range(args.steps):
    torch.cuda.synchronize()
    t = time.time()
    out = tr.train_step(data)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f'step {s}: {dt * 1e3:.1f} ms', {k: round(v, 4) for k, v in out['log_vars'].items()}, model.roi_head.last_stats,
          flush=True)
print('max mem GB', torch.cuda.max_memory_allocated() / 2**30)
