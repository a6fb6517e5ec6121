"""Which tensors of an fp32-parity-mode training step still pay an absmax pass before their plane split (no producer vouched for
their absmax), by call site and size -- the list the plane-emitting / absmax-emitting epilogues have to cover (round 6)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from bonai_amd import kernels as K
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
m.backbone.compute_dtype = torch.float32
sites = collections.Counter()
byt = collections.Counter()
known = [0, 0]
orig = K.split_planes


def spy(x, dtype16):
    if dtype16 == torch.float16 and K._known_amax(x) is None:
        fr = [f for f in traceback.extract_stack()[:-1] if 'bonai_amd' in f.filename][-4:]
        key = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(fr))
        sites[key] += 1
        byt[key] += x.numel() * 4
    else:
        known[0] += 1
        known[1] += x.numel() * 4
    return orig(x, dtype16)


for i in range(3):
    if i == 2:
        K.split_planes = spy
    tr.train_step(data)
torch.cuda.synchronize()
K.split_planes = orig
print(f'# splits with a producer-vouched absmax: {known[0]} ({known[1] / 1e6:.0f} MB); with an absmax pass: {sum(sites.values())} '
      f'({sum(byt.values()) / 1e6:.0f} MB)')
for k, v in sorted(byt.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{v / 1e6:9.1f} MB  n={sites[k]:4d}  {k}')
