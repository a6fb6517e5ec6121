"""Time the fused RoIAlign backward (and the three forwards) on the RoI lists of a real bench step (trained-RPN load; ROI_LIGHT=1:
the random-init RPN's).  A/B of the chunk-pipelined kernel on the fused route: build the library with -DRBM_PIPE_MULTI=1 and pass
it as LOFT_HIP_LIB."""
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
# RoI heads at the trained-RPN load (bench.py's primary loop): the first proposals of every image = jittered gt boxes
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
if not os.environ.get('ROI_LIGHT'):
    m.rpn_head.forward_train = saturated
for _ in range(3):
    tr.train_step(data)
cap = {}
orig = K.roi_align_bwd_multi
def hook(sets, shapes, strides, fs, **k):
    cap['a'] = ([tuple(x.clone() if torch.is_tensor(x) else x for x in s) for s in sets], shapes, strides, fs)
    return orig(sets, shapes, strides, fs, **k)
K.roi_align_bwd_multi = hook
tr.train_step(data)
K.roi_align_bwd_multi = orig
torch.cuda.synchronize()
sets, shapes, strides, fs = cap['a']
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f'bwd multi ({[(s[1].shape[0], s[2], s[3]) for s in sets]}): {timeit(lambda: orig(sets, shapes, strides, fs)):.1f} us')
for s in sets:
    print(f'bwd alone K={s[1].shape[0]} P={s[2]} n_rot={s[3]}: {timeit(lambda: orig([s], shapes, strides, fs)):.1f} us')
feats = [torch.randn(sh[0], sh[2], sh[3], sh[1], device='cuda').bfloat16().permute(0, 3, 1, 2) for sh in shapes]
for s in sets:
    print(f'fwd K={s[1].shape[0]} P={s[2]} n_rot={s[3]}: {timeit(lambda: K.roi_align_fwd(feats, s[1], s[2], strides, fs, s[3])):.1f} us')
