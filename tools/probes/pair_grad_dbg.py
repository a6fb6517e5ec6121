"""Gradients of the 256^2 e2e model with and without the pair fusion: per-parameter relative difference (backbone), and both
against the reference fixture's gradient norms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from bonai_amd.debug import DBG
from bonai_amd.synth import make_batch
import test_e2e_gpu as T
gd = np.load(os.path.join(T.GOLD, 'e2e_256.npz'))
size, batch, num_gt = [int(v) for v in gd['meta']]
res = {}
for tag, off in (('pair', False), ('nopair', True), ('pair2', False)):
    with DBG.override(no_pair_fusion=off):
        m = T._build()
        data = make_batch(batch, size, num_gt, device='cuda')
        out = m.train_step(data)
        out['loss'].backward()
        torch.cuda.synchronize()
        res[tag] = {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
for n in res['pair']:
    if not n.startswith('backbone'):
        continue
    a, b, c = res['pair'][n], res['nopair'][n], res['pair2'][n]
    want = float(gd['gradnorm_' + n]) if 'gradnorm_' + n in gd.files else float('nan')
    print(f'{n:42s} |pair-nopair|/|nopair| {float((a - b).norm() / (b.norm() + 1e-20)):.4f}  |pair-pair2| {float((a - c).norm() / (c.norm() + 1e-20)):.4f}  '
          f'norm pair {float(a.norm()):.4f} nopair {float(b.norm()):.4f} ref {want:.4f}')
