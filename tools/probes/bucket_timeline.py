"""When does each gradient bucket become ready inside the backward pass?  One rank, RCCL group of size 1 with
LOFT_FORCE_REDUCER=1 (the whole hook / side-stream / collective path runs; the collective itself is trivial), HIP events on the
reducer's stream after every bucket's all-reduce.  Prints, per bucket, its size and the time it was released and finished
relative to the start of backward -- what an N-rank run has left to hide is whatever is released after the backward's end.
Ring all-reduce estimate per bucket at 8 ranks: 2 * 7/8 * bytes / (7 links x ~50 GB/s usable per direction ... conservative 300 GB/s)."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOFT_FORCE_REDUCER='1')
dist.init_process_group('nccl', rank=0, world_size=1)
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
red = tr.reducer
marks = []
orig_launch = red._launch
def launch(bi):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())     # producer side: bucket complete
    orig_launch(bi)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(red.stream)                        # collective done
    marks.append((bi, e0, e1))
red._launch = launch
orig_begin = red.begin
t_begin = {}
def begin():
    orig_begin()
    marks.clear()
    t_begin['e'] = torch.cuda.Event(enable_timing=True); t_begin['e'].record(torch.cuda.current_stream())
red.begin = begin
import bonai_amd.kernels as K
orig_sumsq = K.sumsq_
t_end = {}
def sumsq(*a, **k):
    t_end['e'] = torch.cuda.Event(enable_timing=True); t_end['e'].record(torch.cuda.current_stream())
    return orig_sumsq(*a, **k)
K.sumsq_ = sumsq
for it in range(3):
    tr.train_step(data)
    torch.cuda.synchronize()
    if it < 2:
        continue
    total = t_begin['e'].elapsed_time(t_end['e'])
    print(f'reducer.begin -> optimizer (forward + backward + reducer.finish): {total:.2f} ms; {len(red.buckets)} buckets')
    for bi, e0, e1 in marks:
        b = red.buckets[bi]
        mb = (b['end'] - b['start']) * 4 / 2**20
        est = 2 * 7 / 8 * mb * 2**20 / 300e9 * 1e3
        print(f'  bucket {bi:2d}  {mb:6.1f} MiB  {len(b["params"]):3d} params  ready at {t_begin["e"].elapsed_time(e0):6.2f} ms  '
              f'collective done at {t_begin["e"].elapsed_time(e1):6.2f} ms   8-rank ring estimate {est:5.2f} ms')
# ---- which parameter releases each bucket, and when (host enqueue order) its gradient was reported
names = {id(p): n for n, p in m.named_parameters()}
seq = []
orig_hook = red._hook
def hook(p):
    if id(p) not in red._seen:
        ev = torch.cuda.Event(enable_timing=True); ev.record(torch.cuda.current_stream())
        seq.append((id(p), ev))
    orig_hook(p)
red._hook = hook
tr._sink = hook
tr.train_step(data)
torch.cuda.synchronize()
when = {pid: (i, t_begin['e'].elapsed_time(ev)) for i, (pid, ev) in enumerate(seq)}
for bi, b in enumerate(red.buckets):
    ps = sorted(b['params'], key=lambda p: when.get(id(p), (1 << 30, 0))[0])
    ps = [p for p in ps if id(p) in when]
    if not ps:
        print(f'  bucket {bi:2d}: no gradient reported (released by finish())'); continue
    first, last = ps[0], ps[-1]
    print(f'  bucket {bi:2d}: first {names[id(first)]:50s} #{when[id(first)][0]:3d} at {when[id(first)][1]:6.2f} ms | '
          f'last {names[id(last)]:50s} #{when[id(last)][0]:3d} at {when[id(last)][1]:6.2f} ms | unreported {len(b["params"]) - len(ps)}')
