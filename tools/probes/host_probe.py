import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from bonai_amd import nn as F2
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(4):
    tr.train_step(data)
torch.cuda.synchronize()
rows = []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.arena.grad.zero_(); tr.arena.rebind_grads(); tr.reducer.begin()
    for p in tr.arena.params: p._loft_pending = 0
    out = m.train_step(data)
    t1 = time.perf_counter()           # forward enqueued (includes the mid-step host sync)
    F2.GRAD_SINK = tr._sink
    out['loss'].backward()
    F2.GRAD_SINK = None
    t2 = time.perf_counter()           # backward enqueued
    torch.cuda.synchronize()
    t3 = time.perf_counter()           # GPU done
    rows.append((t1 - t0, t2 - t1, t3 - t2))
for r in rows:
    print('fwd(host, incl sync) %.1f ms   bwd enqueue %.1f ms   GPU tail after host done %.1f ms' % tuple(1e3 * x for x in r))
