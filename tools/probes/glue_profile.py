"""GPU time of the NON-conv (aten glue) kernels of one train step, grouped by the python phase that launched them."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
from torch.profiler import record_function
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()


def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function('PH:' + tag):
            return f(*a, **k)
    setattr(obj, name, g)


rpn, roi = model.rpn_head, model.roi_head
wrap(rpn, 'forward_fused', 'rpn.forward'); wrap(rpn, 'loss_fused', 'rpn.loss'); wrap(rpn, 'get_bboxes_fused', 'rpn.proposals')
wrap(roi.bbox_head, 'loss', 'roi.bbox_loss'); wrap(roi.mask_head, 'loss', 'roi.mask_loss'); wrap(roi.offset_head, 'loss', 'roi.offset_loss')
wrap(roi.bbox_head, 'forward', 'roi.bbox_head'); wrap(roi.mask_head, 'forward', 'roi.mask_head'); wrap(roi, '_offset_forward', 'roi.offset_head')
wrap(roi, 'forward_train', 'roi.forward_train'); wrap(model, 'extract_feat', 'backbone+fpn')
tr = Trainer(model)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    with record_function('PH:step'):
        tr.train_step(data)
    torch.cuda.synchronize()
evs = prof.events()
phases = [e for e in evs if e.name.startswith('PH:')]
glue = collections.defaultdict(lambda: [0.0, 0])
names = collections.defaultdict(lambda: [0.0, 0])
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    kt = sum(k.duration for k in e.kernels)
    kn = e.kernels[0].name
    if any(s in kn for s in ('conv_tap', 'conv_wgrad', 'roi_align', 'stem_mfma')):
        continue
    inner = None
    for p in phases:
        if p.thread == e.thread and p.time_range.start <= e.time_range.start and e.time_range.end <= p.time_range.end:
            if inner is None or p.time_range.start >= inner.time_range.start:
                inner = p
    tag = inner.name if inner else ('backward' if e.thread != phases[0].thread else 'other')
    glue[tag][0] += kt; glue[tag][1] += len(e.kernels)
    names[(tag, e.name)][0] += kt; names[(tag, e.name)][1] += len(e.kernels)
for tag, (t, n) in sorted(glue.items(), key=lambda kv: -kv[1][0]):
    print(f'{tag:24s} {t / 1e3:8.3f} ms  {n:4d} kernels')
print()
for (tag, nm), (t, n) in sorted(names.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{tag:22s} {nm:40s} {t / 1e3:8.3f} ms {n:4d}')

print()
big = []
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels or not e.name.startswith('aten::'):
        continue
    kt = sum(k.duration for k in e.kernels)
    if kt >= 8.0:
        big.append((kt, e.name, 'bwd' if e.thread != phases[0].thread else 'fwd', str(e.input_shapes)[:90]))
for kt, nm, th, shp in sorted(big, reverse=True)[:50]:
    print(f'{kt:8.1f} us {th} {nm:24s} {shp}')
