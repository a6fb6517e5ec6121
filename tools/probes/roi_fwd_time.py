"""Time the three RoIAlign forwards of a bench step (RoI lists at the trained-RPN load: first proposals = jittered gt boxes) in
the 16-byte separable kernel (shipped), the 8-byte separable kernel (LOFT_ROI_FWD_SEP4) and the sample-order kernel, and report
the largest difference between them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
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
g = torch.Generator().manual_seed(7)
jit = []
for gb in data['gt_bboxes']:
    b = gb.cpu()
    wh = b[:, 2:] - b[:, :2]
    reps = [(b + (torch.rand(b.shape[0], 4, generator=g) - 0.5) * 0.16 * torch.cat([wh, wh], 1)).clamp(0, 1024) for _ in range(4)]
    jb = torch.cat(reps, 0)
    jit.append(torch.cat([jb, torch.ones(jb.shape[0], 1)], 1))
njit = min(j.shape[0] for j in jit)
jit = torch.stack([j[:njit] for j in jit]).cuda()
orig_ft = m.rpn_head.forward_train
def saturated(*a, **k):
    losses, (props, counts) = orig_ft(*a, **k)
    props = props.clone()
    props[:, :njit] = jit
    return losses, (props, counts.clamp(min=njit))
m.rpn_head.forward_train = saturated
for _ in range(3):
    tr.train_step(data)
calls = []
orig = K.roi_align_fwd
def hook(feats, rois, P, strides, finest_scale=56, n_rot=1):
    calls.append(([f.detach().clone() for f in feats], rois.detach().clone(), P, strides, finest_scale, n_rot))
    return orig(feats, rois, P, strides, finest_scale, n_rot)
K.roi_align_fwd = hook
tr.train_step(data)
K.roi_align_fwd = orig
torch.cuda.synchronize()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot = {}
for feats, rois, P, strides, fs, n_rot in calls:
    wh = (rois[:, 3:5] - rois[:, 1:3])
    lv = K.map_roi_levels(rois, len(feats), fs)
    desc = f'K={rois.shape[0]} P={P} n_rot={n_rot} levels={torch.bincount(lv.long(), minlength=4).tolist()} mean side {wh.mean().item():.0f}px'
    outs = {}
    line = []
    variants = [('shipped', K.ROI_AUTO), ('sep8', K.ROI_FWD_SEP4), ('sample', K.ROI_FWD_SAMPLE), ('stream-only', 255 << 8)]
    variants += [(f'lds{kb}k/s{sp}', (kb << 8) | (sp << 22)) for kb in (24, 40, 255) for sp in (1, 2, 4)]
    for name, var in variants:
        K.ROI_FWD_VARIANT = var
        try:
            t = timeit(lambda: orig(feats, rois, P, strides, fs, n_rot))
            outs[name] = orig(feats, rois, P, strides, fs, n_rot).float()
        finally:
            K.ROI_FWD_VARIANT = K.ROI_AUTO
        tot[name] = tot.get(name, 0.0) + t
        line.append(f'{name} {t:.1f} us')
    d1 = max((outs[n] - outs['sample']).abs().max().item() for n in outs if n not in ('sep8', 'sample'))
    d2 = (outs['sep8'] - outs['sample']).abs().max().item()
    print(desc + ':\n   ' + ', '.join(line) + '\n  ' + f'; max |16-byte forms - sample| {d1:.4g}, |sep8 - sample| {d2:.4g}, range {outs["sample"].abs().max().item():.3g}')
print('per step: ' + ', '.join(f'{k} {v:.1f} us' for k, v in tot.items()))
