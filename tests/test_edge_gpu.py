"""GPU edge cases of the training step: images without any ground truth (no positives anywhere -> the mask and
offset heads see zero RoIs), ragged gt counts across the batch, and the reference's list-of-proposals input form."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    torch.manual_seed(0)
    return build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()


def test_no_gt_and_ragged_gt():
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    m = _model()
    tr = Trainer(m, lr=1e-3)
    data = make_batch(3, 256, 6, device='cuda')
    # image 0: no gt at all; image 1: 2 gts; image 2: 6 gts
    for k in ('gt_bboxes', 'gt_labels', 'gt_masks', 'gt_offsets'):
        data[k][0] = data[k][0][:0]
        data[k][1] = data[k][1][:2]
    out = tr.train_step(data)
    lv = dict(out['log_vars'].items())
    assert all(np.isfinite(v) for v in lv.values()), lv
    # every image empty: zero positives -> mask / offset losses are exactly 0, the step still runs
    for k in ('gt_bboxes', 'gt_labels', 'gt_masks', 'gt_offsets'):
        data[k] = [t[:0] for t in data[k]]
    out = tr.train_step(data)
    lv = dict(out['log_vars'].items())
    assert all(np.isfinite(v) for v in lv.values()), lv
    assert lv['loss_mask'] == 0.0 and lv['loss_offset'] == 0.0 and m.roi_head.last_stats['num_pos'] == 0
    assert torch.isfinite(tr.arena.data).all()


def test_rccl_reducer_path_single_rank():
    """The N>1 machinery (autograd hooks -> ordered buckets -> side-stream RCCL all-reduce -> event wait) on one GPU:
    a 1-rank 'nccl' group with the reducer forced on must reproduce the plain step exactly."""
    import socket
    import torch.distributed as dist
    from bonai_amd.engine import Trainer
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    RandomSampler.choice_mode = 'first'
    try:
        data = make_batch(2, 256, 6, device='cuda')
        m0 = _model()
        ref = dict(Trainer(m0, lr=1e-3).train_step(data)['log_vars'].items())
        p_ref = next(p for p in m0.parameters() if p.requires_grad).detach().clone()
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOFT_FORCE_REDUCER='1')
        dist.init_process_group('nccl', rank=0, world_size=1)
        try:
            m1 = _model()
            tr = Trainer(m1, lr=1e-3, bucket_bytes=32 << 20)
            assert tr.reducer.enabled and len(tr.reducer.buckets) >= 3
            got = dict(tr.train_step(data)['log_vars'].items())
            torch.cuda.synchronize()
            p_got = next(p for p in m1.parameters() if p.requires_grad).detach()
        finally:
            dist.destroy_process_group()
            os.environ.pop('LOFT_FORCE_REDUCER', None)
        for k in ref:
            assert abs(ref[k] - got[k]) <= 2e-3 * max(1.0, abs(ref[k])), (k, ref[k], got[k])   # wgrad atomics reorder fp32 sums
        assert torch.allclose(p_ref, p_got, atol=1e-4)
    finally:
        RandomSampler.choice_mode = 'random'


def test_forward_dummy_shapes():
    """two_stage.py:87-103 / standard_roi_head.py:54-68 contract used by tools/get_flops.py."""
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
    with torch.no_grad():
        (cls, reg), (cls_score, bbox_pred, mask_pred) = m.forward_dummy(torch.randn(1, 3, 256, 256, device='cuda'))
    assert len(cls) == 5 and cls[0].shape[1] == 3 and reg[0].shape[1] == 12 and cls[0].shape[2:] == (64, 64)
    assert cls_score.shape == (1000, 2) and bbox_pred.shape == (1000, 4) and mask_pred.shape[0] == 100 and mask_pred.shape[2:] == (28, 28)


def test_trainer_direct_grad_sink_matches_autograd_accumulation():
    """Trainer path (kernels accumulate weight/BN gradients straight into the flat arena, bonai_amd.nn.GRAD_SINK) vs the
    plain autograd accumulation on the same step: every gradient identical up to the run-to-run noise of the backward itself
    (split-K fp32 atomics; the packed-bf16 atomics of the sparse RPN scatter) -- 1e-2 relative in norm."""
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    data = make_batch(2, 256, 8, device='cuda')

    def build():
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
        return m.cuda().train()
    from bonai_amd.debug import DBG

    def autograd_grads():
        ref = build()
        with DBG.override(no_side_stream=True):          # reference: one stream, plain autograd accumulation
            ref.train_step(data)['loss'].backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}

    def rel(a, b):
        return (a - b).norm().item() / (b.norm().item() + 1e-12)
    want = autograd_grads()
    # (diagnostic, round 3: the driver once saw an uncorrelated layer2.0.conv1 gradient here that 176 repetitions in fresh and
    #  in suite-history processes never reproduced.  Should it come back, the message says WHICH side moved: the reference is
    #  computed twice, and a mismatching Trainer step is repeated once.)
    again = autograd_grads()
    unstable = [(n, rel(again[n], w)) for n, w in want.items() if rel(again[n], w) > 1e-2]
    assert not unstable, ('plain autograd path differs between two runs', unstable[:4])
    m = build()
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    # the trainer path also runs the bbox / mask branches on the side stream (forward and backward) and deposits through the
    # unpack queue; lr = 0 keeps the weights, so every repetition must reproduce the same gradients (a stream race would not)
    for rep in range(3):
        lv = dict(tr.train_step(data, lr=0.0)['log_vars'].items())
        torch.cuda.synchronize()
        got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
        assert set(want) <= set(got)
        bad = [(n, round(rel(got[n], w), 4)) for n, w in want.items() if (got[n] - w).norm().item() > 1e-2 * w.norm().item() + 1e-7]
        if bad:
            first = {n: got[n].clone() for n, _ in bad[:8]}
            tr.train_step(data, lr=0.0)
            torch.cuda.synchronize()
            retry = [(n, round(rel(m.get_parameter(n).grad, want[n]), 4), round(rel(m.get_parameter(n).grad, first[n]), 4)) for n in first]
            raise AssertionError(f'rep {rep}: {len(bad)} of {len(want)} gradients off; first {bad[:6]}; losses {lv}; '
                                 f'the same step repeated (name, vs autograd, vs the failing step): {retry}')
        for n, g in got.items():
            if n not in want:
                assert g.abs().max().item() == 0, n


def test_static_loss_scale_is_transparent():
    """Fp16OptimizerHook semantics (hooks.py:64-96): scaling the loss by 512 and un-scaling inside the fused clip+SGD kernel
    leaves the update unchanged (up to bf16 rounding of the scaled gradients flowing through the backward)."""
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    data = make_batch(1, 256, 6, device='cuda')
    res = []
    for scale in (1.0, 512.0):
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        sd0 = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(sd0)
        m = m.cuda().train()
        tr = Trainer(m, lr=0.01, momentum=0.9, weight_decay=1e-4, max_norm=35.0, loss_scale=scale)
        tr.train_step(data)
        res.append({n: (p.detach().cpu() - sd0[n]) for n, p in m.named_parameters() if p.requires_grad})
    for n in res[0]:
        a, b = res[0][n], res[1][n]
        assert (a - b).norm().item() <= 2e-2 * b.norm().item() + 1e-9, (n, (a - b).norm().item(), b.norm().item())
