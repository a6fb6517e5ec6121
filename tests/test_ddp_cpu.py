"""CPU, world_size 2 over gloo: the N>1 path -- flat arena, bucketed gradient all-reduce fired from autograd
hooks in reverse registration order, and the fused log-var all-reduce of LOFT._parse_losses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from bonai_amd.engine import BucketedAllReduce, FlatArena
        from bonai_amd.loft.detector import LOFT
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 4))
        arena = FlatArena(model)
        red = BucketedAllReduce(arena, bucket_bytes=4096)   # several buckets
        assert red.enabled and len(red.buckets) >= 2
        covered = sorted((b['start'], b['end']) for b in red.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == arena.numel
        assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))
        # parameters are views into the arena
        p0 = next(model.parameters())
        assert p0.data_ptr() >= arena.data.data_ptr()
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(8, 16, generator=g)
        arena.grad.zero_()
        arena.rebind_grads()
        red.begin()
        model(x).pow(2).sum().backward()
        local = arena.grad.clone()      # may already hold reduced buckets; recompute the local grads separately
        red.finish()
        # reference: every rank's local gradient, summed
        ref = torch.zeros_like(arena.grad)
        for r in range(world):
            m2 = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 4))
            m2.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
            xr = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 + r))
            m2(xr).pow(2).sum().backward()
            for p, p2 in zip(arena.order, reversed(list(m2.parameters()))):
                o = arena.offsets[id(p)]
                ref[o:o + p.numel()] += p2.grad.reshape(-1)
        assert torch.allclose(arena.grad, ref, atol=1e-5), (arena.grad - ref).abs().max()
        # fused log-var reduction: mean over ranks of every entry, one collective
        losses = dict(loss_a=torch.tensor(1.0 + rank), loss_b=[torch.tensor(2.0), torch.tensor(3.0 * rank)], acc=torch.tensor(50.0))
        loss, log_vars, vec = LOFT._parse_losses(losses)
        want = [1.5, 2.0 + 1.5, 50.0, 1.5 + 3.5]
        # (the collective is asynchronous: the values are complete when they are read, as train_step's log_vars are)
        from bonai_amd.loft.detector import _LazyLogVars
        got = list(_LazyLogVars(list(log_vars.keys()), vec).values())
        assert torch.allclose(torch.tensor(got), torch.tensor(want)), got
        q.put((rank, 'ok'))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'


def test_step_lr_schedule():
    from bonai_amd.engine import step_lr
    assert abs(step_lr(0.005, 0, 0) - 0.005 * 0.001) < 1e-12
    assert abs(step_lr(0.005, 300, 0) - 0.005) < 1e-12
    assert abs(step_lr(0.005, 150, 0) - 0.005 * (1 - 0.5 * 0.999)) < 1e-12
    assert abs(step_lr(0.005, 10**6, 16) - 0.0005) < 1e-12 and abs(step_lr(0.005, 10**6, 22) - 0.00005) < 1e-12


def test_bucket_layout_covers_the_arena_and_tapers_at_the_end():
    """BucketedAllReduce's layout alone (no process group): contiguous cover of the arena in arena order, every parameter in
    exactly one bucket, no bucket above the cap except a single oversized tensor, and the LAST buckets cut small (what is
    released at the very end of backward cannot be overlapped)."""
    import torch.nn as nn
    from bonai_amd.engine import FlatArena, BucketedAllReduce
    m = nn.Sequential(*[nn.Linear(512, 512) for _ in range(20)], nn.Linear(2048, 2048), *[nn.Linear(512, 512) for _ in range(20)])   # a 16 MiB tensor among 1 MiB ones
    arena = FlatArena(m)
    cap = 4 << 20
    red = BucketedAllReduce(arena, bucket_bytes=cap)
    spans = [(b['start'], b['end']) for b in red.buckets]
    assert spans[0][0] == 0 and spans[-1][1] == arena.numel
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    flat = [id(p) for b in red.buckets for p in b['params']]
    assert flat == [id(p) for p in arena.order]
    for b in red.buckets:
        assert (b['end'] - b['start']) * 4 <= cap or len(b['params']) == 1
    sizes = [(b['end'] - b['start']) * 4 for b in red.buckets]
    assert sizes[-1] <= cap // 3 and sizes[-1] <= sizes[-2] <= sizes[-3] <= cap
    # a model too small for a taper keeps the plain greedy layout
    small = BucketedAllReduce(FlatArena(nn.Sequential(nn.Linear(64, 64), nn.Linear(64, 64))), bucket_bytes=cap)
    assert len(small.buckets) == 1 and small.buckets[0]['end'] == 2 * (64 * 64 + 64)
