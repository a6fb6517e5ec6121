"""Dump the RoI lists of one bench step's fused RoIAlign backward (gpurun_out/rois.npz) and print the (RoI, 8x8 tile) pair
statistics per pyramid level: total pairs, pairs of the busiest tile -- the launch is as long as its busiest tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bonai_amd import kernels as K
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(3):
    tr.train_step(data)
dump = {}
orig = K.roi_align_bwd_multi
def hook(sets, *a, **k):
    for i, s in enumerate(sets):
        dump[f'rois{i}'] = s[1].float().cpu().numpy(); dump[f'P{i}'] = s[2]; dump[f'nrot{i}'] = s[3]
    return orig(sets, *a, **k)
K.roi_align_bwd_multi = hook
tr.train_step(data)
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'rois.npz'), **dump)
strides = [4, 8, 16, 32]
for lvl, st in enumerate(strides):
    n = 1024 // st
    nt = (n + 7) // 8
    cnt = np.zeros((8, nt, nt), np.int64)
    wcnt = np.zeros((8, nt, nt), np.float64)
    for i in range(3):
        r = dump[f'rois{i}']; P = dump[f'P{i}']
        sc = np.sqrt(np.clip(r[:, 3] - r[:, 1], 0, None) * np.clip(r[:, 4] - r[:, 2], 0, None))
        lv = np.clip(np.floor(np.log2(sc / 56 + 1e-6)), 0, 3).astype(int)
        for b, x1, y1, x2, y2 in r[lv == lvl]:
            tx0, tx1 = int(max(x1 / st - 1, 0)) // 8, int(min(x2 / st + 1, n - 1)) // 8
            ty0, ty1 = int(max(y1 / st - 1, 0)) // 8, int(min(y2 / st + 1, n - 1)) // 8
            cnt[int(b), ty0:ty1 + 1, tx0:tx1 + 1] += 1
            wcnt[int(b), ty0:ty1 + 1, tx0:tx1 + 1] += (P / 7.0) ** 2 * dump[f'nrot{i}'] ** 0
    print(f'level {lvl}: tiles {cnt.size}  pairs {cnt.sum()}  mean/tile {cnt.mean():.1f}  max/tile {cnt.max()}  p99 {np.percentile(cnt, 99):.0f}  '
          f'nonempty {np.count_nonzero(cnt)}')
