"""Is the step host-bound?  Per step: host time until train_step returns (everything enqueued; includes the one mid-step
host read), then the wait for the GPU to drain.  A tail near zero means the GPU was waiting for the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfgname = sys.argv[2] if len(sys.argv) > 2 else 'loft_foa_r50_fpn_2x_bonai.py'
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', cfgname))
if cfg.get('fp16'):
    from bonai_amd import lib as L
    L.set_act16(torch.float16)
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
tr = Trainer(m, lr=0.005, loss_scale=(cfg.get('fp16') or {}).get('loss_scale', 1.0), graph_features=graph)
data = make_batch(8, size, 80 if size >= 512 else 8, device='cuda')
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    tr.train_step(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'size {size}{" graph" if graph else ""}: host {1e3 * (t1 - t0):6.1f} ms   GPU tail {1e3 * (t2 - t1):6.1f} ms   step {1e3 * (t2 - t0):6.1f} ms')
# back-to-back (no sync between steps): the steady-state rate
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(10):
    tr.train_step(data)
torch.cuda.synchronize()
print(f'size {size}{" graph" if graph else ""}: back-to-back {1e2 * (time.perf_counter() - t0):6.1f} ms/step')
