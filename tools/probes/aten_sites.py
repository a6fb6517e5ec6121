"""Which lines of the package issue the step's stock aten kernels?  A TorchDispatchMode over one train step records, for every
aten op that touches a device tensor, the innermost bonai_amd frame of the Python stack (ops run by the autograd thread have
none).  Prints ops per step grouped by (site, op)."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
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
torch.cuda.synchronize()
NOKERNEL = ('view', 'reshape', 'expand', 'permute', 'transpose', 'slice', 'select', 'unsqueeze', 'squeeze', 'as_strided', 'detach',
            'alias', 't.default', 'unbind', 'split', 'empty', 'is_', 'size', 'stride', 'numel', 'sym_', 'record_stream', '_unsafe_view',
            'unfold', 'narrow', 'chunk', 'lift_fresh', 'resize_', 'set_', 'dim', 'storage_offset', '_local_scalar_dense', 'flatten')
count = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace('aten.', '')
        if any(name.startswith(n) or name.split('.')[0] == n for n in NOKERNEL):
            return out
        flat, _ = tree_flatten((args, kwargs, out))
        ts = [t for t in flat if isinstance(t, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        site = '(autograd thread / no package frame)'
        for fr in reversed(traceback.extract_stack()):
            if '/bonai_amd/' in fr.filename and 'probes' not in fr.filename:
                site = f'{fr.filename.split("/bonai_amd/")[-1]}:{fr.lineno}'
                break
        shp = next((tuple(t.shape) for t in ts if t.is_cuda), ())
        count[(site, name, shp)] += 1
        return out
with Mode():
    tr.train_step(data)
torch.cuda.synchronize()
bysite = collections.Counter()
for (site, name, shp), n in count.items():
    bysite[site] += n
print(f'{sum(count.values())} kernel-launching aten calls in the step')
for site, n in bysite.most_common(60):
    ops = sorted(((nm, shp, c) for (s_, nm, shp), c in count.items() if s_ == site), key=lambda x: -x[2])
    print(f'{n:4d}  {site:40s} ' + ', '.join(f'{nm}{list(shp)}x{c}' for nm, shp, c in ops[:int(os.environ.get('MAXOPS', '6'))]))
