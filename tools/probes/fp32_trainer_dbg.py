"""Bisect of tests/test_trainer_gpu.py::test_fp32_mode_trainer_shares_operand_planes_across_streams_safely (round 6 debugging)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from bonai_amd import kernels as K
from bonai_amd.debug import DBG
from bonai_amd.engine import Trainer
from bonai_amd.synth import make_batch
from bonai_amd.loft.core import RandomSampler
import test_trainer_gpu as T
RandomSampler.choice_mode = 'first'
data = make_batch(2, 256, 8, device='cuda')


def run(tag, **dbg):
    m0 = T._synth_model(); m0.backbone.compute_dtype = torch.float32
    want = T._autograd_grads(m0, data)
    m = T._synth_model(); m.backbone.compute_dtype = torch.float32
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    with DBG.override(**dbg):
        tr.train_step(data, lr=0.0)
    torch.cuda.synchronize()
    got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    worst = sorted(((float((got[n] - w).norm() / (w.norm() + 1e-12)), n) for n, w in want.items()), reverse=True)[:3]
    print(tag, ' | '.join(f'{n} {e:.2e}' for e, n in worst), flush=True)


run('default')
run('default again')
K.PLANES_DB_FUSED = False
run('no db fusion')
K.PLANES_DB_FUSED = True
K.AMAX_FROM_PRODUCER = False
run('no producer amax')
K.AMAX_FROM_PRODUCER = True
run('narrow_mfma_bwd', narrow_mfma_bwd=True)
run('no_wgrad_stream', no_wgrad_stream=True)
run('no_side_stream', no_side_stream=True)
run('no_leaf_sink', no_leaf_sink=True)

print('--- with synchronisation points')
orig_flush = K.UnpackQueue.flush
def flush_sync(self):
    torch.cuda.synchronize()
    return orig_flush(self)
K.UnpackQueue.flush = flush_sync
run('sync before every flush')
K.UnpackQueue.flush = orig_flush
orig_nhb = K.narrow_head_bwd
def nhb_sync(*a, **k):
    torch.cuda.synchronize()
    r = orig_nhb(*a, **k)
    torch.cuda.synchronize()
    return r
K.narrow_head_bwd = nhb_sync
run('sync around narrow_head_bwd')
K.narrow_head_bwd = orig_nhb
orig_add = K.UnpackQueue.add
def add_sync(self, *a, **k):
    torch.cuda.synchronize()
    return orig_add(self, *a, **k)
K.UnpackQueue.add = add_sync
run('sync before every queue add')
K.UnpackQueue.add = orig_add
import bonai_amd.nn as F2
orig_q = F2._queue_param_grads
def q_log(jobs):
    r = orig_q(jobs)
    print('   _queue_param_grads', [(tuple(j[0].shape), tuple(j[1].shape), j[1].is_contiguous(), None if j[3] is None else tuple(j[3].shape), j[4]) for j in jobs], '->', r,
          'stream', K.L.stream().value)
    return r
F2._queue_param_grads = q_log
run('logged')
