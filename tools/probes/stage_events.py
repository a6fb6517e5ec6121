"""Where a training step's GPU time goes, stage by stage, WITHOUT a profiler attached: HIP events recorded on the main stream at
the python boundaries of the step (and the host clock at the same points: 'lead' = how far the host runs ahead of the GPU there).
Same workload as bench.py's headline line (8 x 1024^2, trained-RPN load).   usage: python stage_events.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from bonai_amd import kernels as K

cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(model, lr=cfg.optimizer.lr, momentum=cfg.optimizer.momentum, weight_decay=cfg.optimizer.weight_decay,
             max_norm=cfg.optimizer_config.grad_clip.max_norm)
data = make_batch(8, 1024, 80, device='cuda')
g = torch.Generator().manual_seed(7)
jit = []
for gb in data['gt_bboxes']:
    b = gb.cpu(); wh = b[:, 2:] - b[:, :2]
    reps = [(b + (torch.rand(b.shape[0], 4, generator=g) - 0.5) * 0.16 * torch.cat([wh, wh], 1)).clamp(0, 1024) for _ in range(4)]
    jb = torch.cat(reps, 0)
    jit.append(torch.cat([jb, torch.ones(jb.shape[0], 1)], 1))
njit = min(j.shape[0] for j in jit)
jit = torch.stack([j[:njit] for j in jit]).cuda()
MARKS = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    MARKS.append((tag, e, time.perf_counter()))


def wrap(obj, name, before=None, after=None):
    f = getattr(obj, name)
    def w(*a, **k):
        if before: mark(before)
        r = f(*a, **k)
        if after: mark(after)
        return r
    setattr(obj, name, w)


orig_ft = model.rpn_head.forward_train
def saturated(*a, **k):
    mark('rpn head in')
    losses, (props, counts) = orig_ft(*a, **k)
    props = props.clone(); props[:, :njit] = jit
    mark('rpn losses + proposals done')
    return losses, (props, counts.clamp(min=njit))
model.rpn_head.forward_train = saturated
wrap(model.rpn_head, 'forward_fused', after='rpn convs done')
wrap(model.roi_head.bbox_assigner, 'assign_batched', before='roi assign in')
wrap(model.roi_head.bbox_sampler, 'sample_batched', after='roi sampled')
wrap(model.roi_head.bbox_head, 'forward', before='bbox head in (RoIAlign queued, counts read)')
wrap(model.roi_head, 'forward_train', after='forward done')
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(6):
    tr.train_step(data)
torch.cuda.synchronize()
rows = []
for s in range(nsteps):
    MARKS.clear()
    mark('step in')
    tr.train_step(data)
    mark('step out (sgd queued)')
    rows.append(list(MARKS))
torch.cuda.synchronize()
# per-step stage durations (GPU) and the host's lead over the GPU at each mark, measured against the first mark of the first step
e0, h0 = rows[0][0][1], rows[0][0][2]
print(f'{"mark":48s} {"GPU since previous mark (us), median over steps":>50s}   host lead at the mark (ms), per step')
tags = [m[0] for m in rows[-1]]
import statistics
for i, tag in enumerate(tags):
    durs, leads = [], []
    for r in rows[1:]:
        if i >= len(r) or r[i][0] != tag: continue
        if i > 0: durs.append(r[i - 1][1].elapsed_time(r[i][1]) * 1e3)
        leads.append(e0.elapsed_time(r[i][1]) - (r[i][2] - h0) * 1e3)
    d = f'{statistics.median(durs):10.1f}' if durs else ' ' * 10
    print(f'{tag:48s} {d:>50s}   ' + ' '.join(f'{v:6.2f}' for v in leads))
tot = [r[0][1].elapsed_time(r[-1][1]) for r in rows[1:]]
print('step GPU time between the first and last mark (ms):', ' '.join(f'{v:.2f}' for v in tot))
