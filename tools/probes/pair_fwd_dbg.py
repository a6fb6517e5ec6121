import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from bonai_amd.debug import DBG
from bonai_amd.synth import make_batch
import test_e2e_gpu as T
gd = np.load(os.path.join(T.GOLD, 'e2e_256.npz'))
size, batch, num_gt = [int(v) for v in gd['meta']]
outs = {}
for tag, off in (('pair', False), ('nopair', True)):
    with DBG.override(no_pair_fusion=off):
        m = T._build()
        data = make_batch(batch, size, num_gt, device='cuda')
        rec = {}
        hs = []
        for name, mod in m.backbone.named_modules():
            if name.count('.') == 1 and name.startswith('layer'):
                hs.append(mod.register_forward_hook(lambda mod, i, o, name=name: rec.__setitem__(name, o.detach().float().clone())))
        feats = m.extract_feat(data['img'])
        torch.cuda.synchronize()
        outs[tag] = rec
for n in outs['pair']:
    a, b = outs['pair'][n], outs['nopair'][n]
    print(f'{n:12s} rel diff {float((a - b).norm() / (b.norm() + 1e-20)):.5f}  max {float((a - b).abs().max()):.4f}  absmean {float(b.abs().mean()):.4f}')
