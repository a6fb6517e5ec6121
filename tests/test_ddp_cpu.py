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


# ---- world 8 (the node size the driver scales to): bucket ORDER under rank-dependent graphs ------------------------------------
class _Branchy(torch.nn.Module):
    """A trunk and four heads; which heads take part in a rank's loss depends on the rank -- what a rank without positive RoIs
    does to the mask / FOA branches (roi head: no positives -> no mask / offset loss; mmdet/models/roi_heads/loft_roi_head.py)."""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(16, 96), nn.ReLU(), nn.Linear(96, 96), nn.ReLU())
        self.heads = nn.ModuleList([nn.Sequential(nn.Linear(96, 64), nn.ReLU(), nn.Linear(64, 8)) for _ in range(4)])

    def forward(self, x, use):
        f = self.trunk(x)
        return sum(self.heads[i](f).pow(2).sum() * (i + 1) for i in use)


def _heads_of(rank, step):
    # rank 0: every head; rank 3: none at all (its backward never reaches any head); others: rank- and step-dependent subsets,
    # enumerated in rank-dependent ORDER so that the heads' gradients are produced in different sequences on different ranks
    if rank == 0:
        return [0, 1, 2, 3]
    if rank == 3:
        return []
    sel = [i for i in range(4) if ((rank * 5 + step * 3 + i) % 3) != 0]
    return sel[::-1] if rank % 2 else sel


def _worker8(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from bonai_amd.engine import BucketedAllReduce, FlatArena
        torch.manual_seed(0)
        model = _Branchy()
        arena = FlatArena(model)
        red = BucketedAllReduce(arena, bucket_bytes=8192)
        assert red.enabled and len(red.buckets) >= 6
        launched = []
        orig = red._launch
        red._launch = lambda bi: (launched.append(bi), orig(bi))[1]
        for step in range(3):
            use = _heads_of(rank, step)
            x = torch.randn(4, 16, generator=torch.Generator().manual_seed(1000 * step + rank))
            arena.grad.zero_()
            arena.rebind_grads()
            launched.clear()
            red.begin()
            loss = model(x, use) if use else model.trunk(x).sum() * 0.0 + model.trunk[0].weight.sum() * 0.0
            loss.backward()
            red.finish()
            # every rank issued every bucket exactly once, in index order (the collectives pair up by construction)
            assert launched == list(range(len(red.buckets))), launched
            ref = torch.zeros_like(arena.grad)
            for r in range(world):
                m2 = _Branchy()
                m2.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
                u2 = _heads_of(r, step)
                if not u2:
                    continue
                xr = torch.randn(4, 16, generator=torch.Generator().manual_seed(1000 * step + r))
                m2(xr, u2).backward()
                for p, p2 in zip(arena.order, reversed(list(m2.parameters()))):
                    if p2.grad is not None:
                        o = arena.offsets[id(p)]
                        ref[o:o + p.numel()] += p2.grad.reshape(-1)
            assert torch.allclose(arena.grad, ref, rtol=1e-4, atol=1e-4), float((arena.grad - ref).abs().max())
            # bit-identical on every rank (all-reduce result), so the SGD that follows keeps the replicas identical
            mine = arena.grad.clone()
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            assert all(torch.equal(gathered[0], t) for t in gathered)
            with torch.no_grad():
                arena.data -= 1e-3 * arena.grad / world
        q.put((rank, 'ok'))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bucketed_allreduce_gloo_world8_rank_dependent_graphs():
    """eight ranks whose autograd graphs differ (heads missing on some ranks, one rank with no head at
    all, heads finishing in different orders): every rank launches every bucket once, in index order, so the collectives pair up;
    sums equal the sum of the ranks' local gradients; replicas stay bit-identical over three steps."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'
