"""GPU, world_size 2: the REAL LOFT training step under data parallelism on the one available GPU.

Both ranks share cuda:0 and talk over gloo (RCCL cannot put two ranks on one device); everything else is the production
N>1 path: flat arena, kernels depositing gradients straight into it (GRAD_SINK) and releasing buckets, bucketed all-reduce
on the side stream from autograd hooks, fused log-var all-reduce, clip + SGD with 1/world scaling.  Checks: the ranks see
different data, finish with bit-identical parameters, the averaged gradient equals the mean of the ranks' local gradients
(recomputed without the reducer), and the step changes the weights."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, mode='balanced', backend='gloo'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank if backend == 'nccl' else 0)      # 'nccl' (= RCCL): one device per rank; gloo: both on cuda:0
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from bonai_amd.config import Config
        from bonai_amd.engine import Trainer
        from bonai_amd.loft import build_detector
        from bonai_amd.loft.core import RandomSampler
        from bonai_amd.synth import make_batch
        RandomSampler.choice_mode = 'first'
        cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))

        def build():
            torch.manual_seed(0)
            return build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        data = make_batch(1, 256, 6, rank=rank, device='cuda')
        if mode == 'imbalanced':
            # rank 0: an image with no gt at all (no positive RoI -> its graph has no mask / FOA branch, those parameters get
            # no gradient there); rank 1: a crowded image.  The collectives must still pair up bucket by bucket.
            data = make_batch(1, 256, 40 if rank else 6, rank=rank, device='cuda')
            if rank == 0:
                for k in ('gt_bboxes', 'gt_labels', 'gt_masks', 'gt_offsets'):
                    data[k] = [t[:0] for t in data[k]]
        # local gradient of this rank, no reducer (plain autograd)
        ref = build()
        ref.train_step(data)['loss'].backward()
        local = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        del ref
        m = build()
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        tr = Trainer(m, lr=0.01, momentum=0.0, weight_decay=0.0, max_norm=0.0, **({} if mode == 'imbalanced' else dict(bucket_bytes=16 << 20)))
        assert tr.reducer.enabled and tr.reducer.on_gpu and len(tr.reducer.buckets) >= 4 and tr.world == 2
        out = tr.train_step(data)
        torch.cuda.synchronize()
        if mode == 'imbalanced':
            npos = [torch.zeros(1, device='cuda') for _ in range(world)]
            dist.all_gather(npos, torch.tensor([float(m.roi_head.last_stats['num_pos'])], device='cuda'))
            assert npos[0].item() == 0 and npos[1].item() >= 40, npos
            gm = local.get('roi_head.mask_head.convs.0.conv.weight')
            assert (gm is not None and gm.abs().sum() > 0) if rank == 1 else (gm is None or gm.abs().sum() == 0)
        # (1) the summed gradient in the arena equals the sum of both ranks' local gradients
        names = [n for n, p in m.named_parameters() if p.requires_grad]
        for n, p in m.named_parameters():
            if not p.requires_grad:
                continue
            mine = local.get(n, torch.zeros_like(p))
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            want = both[0] + both[1]
            got = p.grad
            assert (got - want).norm().item() <= 2e-2 * want.norm().item() + 1e-7, (n, (got - want).norm().item(), want.norm().item())
        # (2) parameters identical on both ranks after the step, and changed
        for n, p in m.named_parameters():
            both = [torch.zeros_like(p) for _ in range(world)]
            dist.all_gather(both, p.detach())
            assert torch.equal(both[0], both[1]), n
        moved = sum(float((p.detach() - before[n]).abs().sum()) for n, p in m.named_parameters() if p.requires_grad)
        assert moved > 0
        # (3) the logged losses are the mean over ranks (fused all-reduce) -> identical on both ranks
        lv = torch.tensor([float(v) for v in out['log_vars'].values()], device='cuda')
        both = [torch.zeros_like(lv) for _ in range(world)]
        dist.all_gather(both, lv)
        assert torch.equal(both[0], both[1])
        # (4) three more steps: the reducer prunes the autograd hooks of sink-served parameters after two steps (round 6) -- a
        # rank-dependent state (rank 0 of the imbalanced pair never sees the mask / FOA parameters reported and keeps re-arming
        # them) that must not change the order or the content of the collectives: parameters stay bit-identical across ranks
        armed0 = tr.reducer.hooks_armed()
        for _ in range(3):
            tr.train_step(data)
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            both = [torch.zeros_like(p) for _ in range(world)]
            dist.all_gather(both, p.detach())
            assert torch.equal(both[0], both[1]), ('after pruning', n)
        assert tr.reducer.hooks_armed() < armed0, (tr.reducer.hooks_armed(), armed0)
        q.put((rank, 'ok', len(names)))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_loft_trainer_two_ranks_one_gpu():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg, _ in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


@pytest.mark.timeout(900)
def test_loft_trainer_two_ranks_imbalanced():
    """One rank without a single gt box, the other with 40: no deadlock, the mask / FOA gradients are rank 1's alone, the
    parameters stay bit-identical (default 25 MiB buckets)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 'imbalanced')) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg, _ in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def _prune_worker(port, q):
    """One rank, RCCL, reducer forced on (the mode of bench.py's comm_forced_1rank leg): hook pruning across steps."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOFT_FORCE_REDUCER'] = '1'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from bonai_amd.config import Config
        from bonai_amd.debug import DBG
        from bonai_amd.engine import Trainer
        from bonai_amd.loft import build_detector
        from bonai_amd.loft.core import RandomSampler
        from bonai_amd.synth import make_batch
        RandomSampler.choice_mode = 'first'
        cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))

        def build():
            torch.manual_seed(0)
            return build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        data = make_batch(2, 256, 8, device='cuda')
        ref = build()
        with DBG.override(no_side_stream=True):
            ref.train_step(data)['loss'].backward()
        want = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        del ref
        m = build()
        tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0, max_norm=0.0)
        red = tr.reducer
        assert red.enabled and red.on_gpu and red.prune_hooks
        n_all = len(tr.arena.order)
        assert red.hooks_armed() == n_all

        def check(tag):
            torch.cuda.synchronize()
            for n, p in m.named_parameters():
                if p.requires_grad and n in want:
                    d, s_ = (p.grad - want[n]).norm().item(), want[n].norm().item()
                    assert d <= 1.5e-2 * s_ + 1e-7, (tag, n, d, s_)
        armed = []
        for it in range(4):
            tr.train_step(data, lr=0.0)
            check(f'step {it}')
            armed.append(red.hooks_armed())
        # almost every gradient is deposited by a kernel and reported through the sink: those hooks are gone after two steps
        assert armed[0] == n_all and armed[-1] <= n_all // 4 and armed[-1] == armed[-2], armed
        # every bucket was still released and reduced, in index order, by the sink reports alone
        assert red._next == len(red.buckets)
        # the sink switched off: every gradient comes through autograd's accumulation, no hook fires for the pruned ones -> their
        # buckets leave from finish(); gradients unchanged, and the hooks are re-armed for the next step
        with DBG.override(no_grad_sink=True):
            tr.train_step(data, lr=0.0)
            check('no_grad_sink after pruning')
            assert red.hooks_armed() == n_all, (red.hooks_armed(), n_all)
            tr.train_step(data, lr=0.0)
            check('no_grad_sink, hooks re-armed')
        for it in range(3):
            tr.train_step(data, lr=0.0)
            check(f'sink back, step {it}')
        assert red.hooks_armed() == armed[-1], (red.hooks_armed(), armed)
        q.put((0, 'ok', armed))
    except Exception:  # noqa
        import traceback
        q.put((0, traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_reducer_prunes_autograd_hooks_of_sink_served_parameters():
    """the reducer's host path.  ~300 python post-accumulate hooks per step did nothing but return for the
    parameters whose gradients the kernels deposit themselves; they are removed after two sink-served steps, re-armed when a
    parameter stops reporting, and the reduced gradients equal plain autograd's throughout (one-rank RCCL group, reducer forced)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_prune_worker, args=(_free_port(), q))
    p.start()
    rank, msg, armed = q.get(timeout=800)
    p.join(60)
    assert msg == 'ok', msg
    print('autograd hooks armed per step:', armed)


def _run_two(mode, backend):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg, _ in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank: runs on a node with >= 2 GPUs')


@needs_two_gpus
@pytest.mark.timeout(900)
@pytest.mark.parametrize('mode', ['balanced', 'imbalanced'])
def test_loft_trainer_two_ranks_rccl(mode):
    """the same two-rank checks over the REAL transport -- backend 'nccl' (RCCL over xGMI), one device
    per rank, collectives on the reducer's side stream.  Skipped on the 1-GPU boxes gpurun hands out; it runs the day the suite
    sees two devices (summed gradients = sum of the ranks' local ones, parameters bit-identical on both ranks, mean log-vars)."""
    _run_two(mode, 'nccl')


@needs_two_gpus
@pytest.mark.timeout(900)
def test_bench_launch_contract_two_ranks_rccl():
    """The driver's command line with N = 2 over RCCL (no LOFT_BENCH_SHARED_GPU): one JSON line, 2 ranks seen through 'nccl'."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('LOFT_BENCH_SHARED_GPU', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '2', '--size', '512', '--num-gt', '20']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['comm']['backend'] == 'rccl' and j['comm']['rccl_ranks_seen'] == 2 and j['value'] > 0


@pytest.mark.timeout(900)
def test_bench_launch_contract_two_ranks():
    """The driver's multi-GPU command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...) end to end with N = 2 ranks sharing the one GPU (gloo instead of RCCL):
    exactly one JSON line from rank 0, whole-job value, weak scaling fields."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, LOFT_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '2', '--size', '512', '--num-gt', '20']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['steps'] == 2 and j['warmup'] == 1
    assert j['config']['global_batch'] == 4 and j['config']['parallelism'] == 'dp2' and j['value'] > 0
    assert 'roofline' in j and 'cpu_baseline' not in j
    assert j['comm']['rccl_ranks_seen'] == 2 and j['comm']['buckets'] >= 8 and j['comm']['bucket_mib'] <= 52
    assert 'exposed_ms' in j['comm']


@pytest.mark.timeout(1500)
def test_bench_launch_contract_eight_ranks():
    """the driver's N = 8 command line on this 1-GPU box -- eight ranks sharing the device over gloo at a
    tiny size: the rendezvous, the reducer's bucket order under eight autograd threads' worth of rank skew, the max-over-ranks
    timing and the single JSON line with the whole-job value all run before the driver's first real 8-GPU launch does."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, LOFT_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
           '--batch', '1', '--size', '256', '--num-gt', '8', '--no-light', '--no-roofline']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 8 and j['scaling'] == 'weak' and j['steps'] == 2 and j['warmup'] == 1
    assert j['config']['global_batch'] == 8 and j['config']['parallelism'] == 'dp8' and j['value'] > 0
    assert j['comm']['rccl_ranks_seen'] == 8 and 'cpu_baseline' not in j and j.get('value_fp32_parity') is None
