"""Who issues host->device copies during a train step (each one is a ~4 us bubble on the GPU queue)."""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd import kernels as K
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(model)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(2):
    tr.train_step(data)
cnt = collections.Counter()
def site():
    st = traceback.extract_stack(limit=8)[:-2]
    return ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st) if 'bonai_amd' in f.filename or 'bench' in f.filename)[:150]
orig_h2d = K.h2d
def h2d(values, dtype, device):
    cnt[('h2d', site())] += 1
    return orig_h2d(values, dtype, device)
K.h2d = h2d
orig_tensor = torch.tensor
def tensor(*a, **k):
    dev = k.get('device')
    if dev is not None and 'cuda' in str(dev):
        cnt[('torch.tensor', site())] += 1
    return orig_tensor(*a, **k)
torch.tensor = tensor
orig_to = torch.Tensor.to
def to(self, *a, **k):
    if not self.is_cuda and (any('cuda' in str(x) for x in a) or 'cuda' in str(k.get('device', ''))):
        cnt[('to', site())] += 1
    return orig_to(self, *a, **k)
torch.Tensor.to = to
orig_cuda = torch.Tensor.cuda
def cuda(self, *a, **k):
    if not self.is_cuda:
        cnt[('cuda', site())] += 1
    return orig_cuda(self, *a, **k)
torch.Tensor.cuda = cuda
tr.train_step(data)
torch.cuda.synchronize()
for (kind, s), n in cnt.most_common(40):
    print(n, kind, s)
print('total', sum(cnt.values()))
