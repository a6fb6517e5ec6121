"""Per-parameter gradient-norm error of the fp32 parity mode against the reference fixture (tests/golden/e2e_256.npz), for both
contractions (LOFT_F32_SPLIT6 / SPLIT3 / EXACT): the distribution behind test_e2e_fp32_parity_mode_vs_reference_fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from bonai_amd import kernels as K
from bonai_amd.synth import make_batch
from test_e2e_gpu import _build
gd = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_256.npz'))
size, batch, num_gt = [int(v) for v in gd['meta']]
data = make_batch(batch, size, num_gt, device='cuda')
names = [k[len('allnorm_'):] for k in gd.files if k.startswith('allnorm_')]
res = {}
for cname, c in (('split6', K.F32_SPLIT6), ('split3', K.F32_SPLIT3), ('exact', K.F32_EXACT)):
    K.F32_CONTRACT = c
    m = _build(); m.backbone.compute_dtype = torch.float32
    out = m.train_step(data); out['loss'].backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    res[cname] = {n: abs(float(grads[n].float().norm()) - float(gd['allnorm_' + n])) / max(float(gd['allnorm_' + n]), 1e-12) for n in names}
    lv = dict(out['log_vars'].items())
    print(cname, 'losses', {k: round(float(v), 6) for k, v in lv.items()})
K.F32_CONTRACT = K.F32_SPLIT6
for cname in res:
    e = np.array(sorted(res[cname].values()))
    print(f'{cname}: grad-norm rel err  median {np.median(e):.2e}  p90 {np.percentile(e, 90):.2e}  p99 {np.percentile(e, 99):.2e}  max {e.max():.2e}')
worst = sorted(res['split3'].items(), key=lambda kv: -kv[1])[:15]
for n, v in worst:
    print(f'  {n:60s} split3 {v:.2e}   split6 {res["split6"][n]:.2e}   exact {res["exact"][n]:.2e}')
