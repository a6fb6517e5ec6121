"""Is the fused bottleneck tail at least as close to the fp32 parity path as the unfused launches?  Backbone stage outputs of the
e2e test model (random synthetic weights) at the odd-size cases, bf16 fused / bf16 unfused against fp32; block outputs through
forward hooks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from bonai_amd.debug import DBG
from bonai_amd.synth import make_batch
from test_e2e_gpu import _build
m = _build()
outs = {}
def hook(name):
    def f(mod, inp, out):
        outs[name] = out.detach().float().clone()
    return f
for i, blk in enumerate(m.backbone.layer1):
    blk.register_forward_hook(hook(f'layer1.{i}'))
for i, blk in enumerate(m.backbone.layer2):
    blk.register_forward_hook(hook(f'layer2.{i}'))
for size, batch in ((320, 3), (384, 2), (256, 2)):
    data = make_batch(batch, size, 9, device='cuda')
    res = {}
    for mode, dt, fus in (('f32', torch.float32, True), ('fused', None, True), ('unfused', None, False)):
        with DBG.override(no_bneck_fusion=not fus), torch.no_grad():
            m.backbone.compute_dtype = dt
            outs.clear()
            feats = m.extract_feat(data['img'])
            res[mode] = dict(outs)
            res[mode].update({f'fpn{i}': f.detach().float().clone() for i, f in enumerate(feats)})
    line = []
    for k in res['f32']:
        ref = res['f32'][k]
        ef = float((res['fused'][k] - ref).norm() / ref.norm()); eu = float((res['unfused'][k] - ref).norm() / ref.norm())
        mx = float((res['fused'][k] - res['unfused'][k]).abs().max())
        line.append(f'{k}: fused {ef:.2e} unfused {eu:.2e} (max |f-u| {mx:.3g})')
    print(size, batch, '\n   ' + '\n   '.join(line))
