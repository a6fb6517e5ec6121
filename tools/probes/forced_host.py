"""Host-side cost of the reducer on the forced one-rank RCCL leg: enqueue time of a train step (no synchronisation inside) plain vs
forced, cost of one c10d all_reduce call on a bucket-sized view, and the step time.  MODE=plain|forced python tools/probes/forced_host.py"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
mode = os.environ.get('MODE', 'plain')
if mode == 'forced':
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOFT_FORCE_REDUCER='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
from bonai_amd.config import Config
from bonai_amd.engine import Trainer
from bonai_amd.loft import build_detector
from bonai_amd.synth import make_batch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
torch.manual_seed(0)
m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
tr = Trainer(m, lr=0.005)
data = make_batch(8, 1024, 80, device='cuda')
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
enq, wall = [], []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        a = time.perf_counter(); tr.train_step(data); enq.append(time.perf_counter() - a)
    torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) / 5)
print(f'{mode}: step {min(wall) * 1e3:.2f} ms (min of 3 x 5), host enqueue per step {sorted(enq)[len(enq) // 2] * 1e3:.2f} ms median, min {min(enq) * 1e3:.2f}')
if mode == 'forced':
    red = tr.reducer
    v = tr.arena.grad[:25 << 18]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ws = [dist.all_reduce(v, async_op=True) for _ in range(50)]
    t1 = time.perf_counter()
    ws[-1].wait()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'c10d all_reduce(async) host cost {(t1 - t0) / 50 * 1e6:.1f} us per call; wait() {(t2 - t1) * 1e6:.1f} us; buckets {len(red.buckets)}')
    # device-side cost of the event chain of one bucket: 15 x (event on main, side stream waits, all_reduce from side, main waits)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    s.record()
    for _ in range(15):
        w = dist.all_reduce(v, async_op=True)
    w.wait()
    e.record()
    torch.cuda.synchronize()
    print(f'device time of 15 back-to-back one-rank all_reduce + one wait on an idle device: {s.elapsed_time(e) * 1e3:.0f} us')
    dist.destroy_process_group()
