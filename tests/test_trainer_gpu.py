"""GPU, composite: the Trainer path that bench.py times (direct gradient sink + unpack queue + weight-gradient side stream +
prepack + zero / scratch pools + feature hub) against plain autograd, under every A/B switch of bonai_amd/debug.py; the
under-filled-sampler case of the speculative bbox RoIAlign; checkpoint resume against an uninterrupted run.
Reference for what the Trainer must equal: mmcv's OptimizerHook + torch.optim.SGD as driven from mmdet/apis/train.py:84-108."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    from bonai_amd.config import Config
    return Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))


def _synth_model(cfg=None):
    from bonai_amd.loft import build_detector
    from oracle.synth_weights import synth_tensor
    cfg = cfg or _cfg()
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    return m.cuda().train()


def _autograd_grads(m, data, **switches):
    from bonai_amd.debug import DBG
    with DBG.override(**dict({k: False for k in DBG.active()}, no_side_stream=True, **switches)):
        m.train_step(data)['loss'].backward()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


def _compare(want, m, tag, tol=1.5e-2):
    # (run-to-run noise of ONE configuration, 126 fresh-process runs: worst 6.4e-3, a BN gamma gradient of layer2 --
    #  profiles/round3_probes/trainer_race.txt; a wrong path is off by O(1))
    got = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    assert set(want) <= set(got)
    bad = []
    for n, w in want.items():
        d, s = (got[n] - w).norm().item(), w.norm().item()
        if d > tol * s + 1e-7:
            bad.append((n, d, s))
    assert not bad, (tag, len(bad), bad[:5])
    for n, g in got.items():
        if n not in want:
            assert g.abs().max().item() == 0, (tag, n)


@pytest.fixture()
def first_k():
    from bonai_amd.loft.core import RandomSampler
    RandomSampler.choice_mode = 'first'
    yield
    RandomSampler.choice_mode = 'random'


def test_every_debug_switch_alone_matches_the_default_step(first_k):
    """One Trainer, one batch, lr = 0: the default configuration and then each switch of bonai_amd.debug.SWITCHES alone must
    produce the gradients of plain autograd (1e-2 in norm: split-K fp32 atomics and the packed-bf16 atomics of the sparse RPN
    scatter reorder sums from run to run).  no_block_fusion / no_linear_fn change the autograd graph, so they get their own model."""
    from bonai_amd.debug import DBG, SWITCHES
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    data = make_batch(2, 256, 8, device='cuda')
    want = _autograd_grads(_synth_model(), data)
    # (no_pair_fusion, and no_block_fusion which implies it: the separate launches differ from the fused pair in ~1 element of 5e5 by
    #  one bf16 unit, enough to flip one proposal selection of this random-weight detector and move single gradients by percents --
    #  those two switches are compared with plain autograd on THEIR arithmetic)
    want_sep = _autograd_grads(_synth_model(), data, no_pair_fusion=True)
    m = _synth_model()
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    for rep in range(2):
        tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
        _compare(want, m, f'default rep {rep}')
    assert set(SWITCHES) == set(DBG.__slots__)
    for sw in SWITCHES:
        with DBG.override(**{sw: True}):
            assert DBG.active() == [sw]
            tr.train_step(data, lr=0.0)
            torch.cuda.synchronize()
            _compare(want_sep if sw in ('no_pair_fusion', 'no_block_fusion') else want, m, sw)
    assert DBG.active() == []
    tr.train_step(data, lr=0.0)                       # and back: the default path is intact after every detour
    torch.cuda.synchronize()
    _compare(want, m, 'default after the switches')


def test_fp32_mode_trainer_shares_operand_planes_across_streams_safely(first_k):
    """in the fp32 parity mode a residual block's backward splits its output gradient into operand planes ONCE, on the
    weight-gradient side stream, and the data gradient on the main stream takes them from the memo; the forward's cached planes are
    made on main and consumed (and freed) by the weight gradient on the side stream.  Both hand-overs now carry an event and a
    record_stream (kernels._PlanesEntry): the Trainer step with the side stream must equal the one without it (and plain autograd),
    repeatedly, and the cross-stream path must actually have been taken."""
    from bonai_amd import kernels as K
    from bonai_amd.debug import DBG
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    data = make_batch(2, 256, 8, device='cuda')
    m0 = _synth_model()
    m0.backbone.compute_dtype = torch.float32
    want = _autograd_grads(m0, data)
    m = _synth_model()
    m.backbone.compute_dtype = torch.float32
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    K.PLANES_XSTREAM['waits'] = 0
    for rep in range(3):
        tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
        _compare(want, m, f'fp32 side-stream rep {rep}', tol=1e-4)      # (measured 1.3e-6; the same-slot race of round 6 was 8-30 %)
    assert K.PLANES_XSTREAM['waits'] > 0, 'no operand planes crossed streams: the test does not exercise the hand-over'
    with DBG.override(no_wgrad_stream=True):
        n0 = K.PLANES_XSTREAM['waits']
        tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
        _compare(want, m, 'fp32 no_wgrad_stream', tol=1e-4)
        assert K.PLANES_XSTREAM['waits'] == n0


def test_underfilled_sampler_drops_the_speculative_roialign_cleanly(first_k):
    """with few proposals the sampler cannot fill its 1024 slots per image, the bbox features computed
    speculatively on the worst-case list are dropped -- the fused RoIAlign backward must still launch when the THREE real lists
    have arrived (not from a fallback flush), and the Trainer's gradients must equal autograd's."""
    from bonai_amd import nn as F2
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    cfg = _cfg()
    for k in ('nms_post', 'max_num'):
        cfg.train_cfg.rpn_proposal[k] = 200           # << the RCNN sampler's num = 1024 -> M != B * num
    data = make_batch(2, 256, 8, device='cuda')
    want = _autograd_grads(_synth_model(cfg), data)
    m = _synth_model(cfg)
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
    seen = []
    flush = F2._hub_flush

    def spy():
        if F2.HUB is not None and F2.HUB.get('_pending'):
            seen.append((F2.HUB['_seen'], F2.HUB['_expected'], len(F2.HUB['_pending'])))
        return flush()
    F2._hub_flush = spy
    try:
        tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
    finally:
        F2._hub_flush = flush
    assert m.roi_head.last_stats['num_rois'] < 2 * 1024           # the case under test really is under-filled
    assert seen == [(3, 3, 3)], seen                               # one flush, by the last of exactly three expected lists
    _compare(want, m, 'under-filled')


def test_resume_equals_uninterrupted(tmp_path, first_k):
    """ADVICE r2 (engine.py): 2 steps + save + (fresh model, load, load_optimizer_state) + 2 steps == 4 uninterrupted steps,
    momentum and weight decay on.  Compared: the logged losses of every step and the UPDATE every parameter received over the
    four steps (final - initial), against the same quantities of a SECOND uninterrupted run (the noise floor: atomics order,
    an NMS decision that flips) -- the resumed run must sit within 3x that floor (+ a small absolute term); a resume that loses
    the momentum buffers is shown to sit far outside (control).  The sampler's call counter (its random draws are a function of
    torch.initial_seed() and that count) must come back from the checkpoint and advance as in the uninterrupted run."""
    from bonai_amd import kernels as K
    from bonai_amd.checkpoint import load_checkpoint, save_checkpoint
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    batches = [make_batch(2, 256, 8, step=s, device='cuda') for s in range(4)]
    kw = dict(lr=2e-3, momentum=0.9, weight_decay=1e-4, max_norm=35.0)

    def run(tr, steps):
        return [dict(tr.train_step(batches[s])['log_vars'].items()) for s in steps]
    init = {n: p.detach().clone() for n, p in _synth_model().named_parameters()}

    def update_distance(ma, mb):
        pa, num, den = dict(ma.named_parameters()), 0.0, 0.0
        for n, p in mb.named_parameters():
            if p.requires_grad:
                ua, ub = pa[n].detach() - init[n], p.detach() - init[n]
                num += float((ua - ub).pow(2).sum())
                den += float(ua.pow(2).sum())
        return (num / den) ** 0.5
    K._SAMPLE_CALLS[0] = 0
    a = Trainer(_synth_model(), **kw)
    logs_a = run(a, range(4))
    calls_a = K._SAMPLE_CALLS[0]
    K._SAMPLE_CALLS[0] = 0
    a2 = Trainer(_synth_model(), **kw)                                 # the noise floor: the same four steps once more
    logs_a2 = run(a2, range(4))
    K._SAMPLE_CALLS[0] = 0
    b = Trainer(_synth_model(), **kw)
    run(b, range(2))
    f = str(tmp_path / 'latest.pth')
    save_checkpoint(b.model, f, optimizer_state=b.optimizer_state_dict(), meta=dict(iter=2))
    calls_b = K._SAMPLE_CALLS[0]
    del b
    K._SAMPLE_CALLS[0] = 12345                                         # whatever another process would have
    m = _synth_model()
    c = Trainer(m, **kw)
    ck = load_checkpoint(m, f, strict=True)
    c.load_optimizer_state(ck['optimizer'])
    assert c.iter == 2 and K._SAMPLE_CALLS[0] == calls_b == ck['optimizer']['sampler_calls']
    logs_c = run(c, range(2, 4))
    assert K._SAMPLE_CALLS[0] == calls_a                               # the counter advanced exactly as in the uninterrupted run
    torch.cuda.synchronize()
    floor_u = update_distance(a.model, a2.model)
    floor_m = (a.arena.momentum - a2.arena.momentum).norm().item() / a.arena.momentum.norm().item()
    d_u = update_distance(a.model, m)
    d_m = (a.arena.momentum - c.arena.momentum).norm().item() / a.arena.momentum.norm().item()
    print(f'resume: update distance {d_u:.3e} (floor {floor_u:.3e}), momentum distance {d_m:.3e} (floor {floor_m:.3e})')
    assert d_u <= 3 * floor_u + 2e-2 and d_m <= 3 * floor_m + 2e-2, (d_u, floor_u, d_m, floor_m)
    for la, l2, lc in zip(logs_a[2:], logs_a2[2:], logs_c):
        for k in la:
            tol = 3 * abs(la[k] - l2[k]) + (3.0 if k == 'acc' else 8e-2 * max(1.0, abs(la[k])))
            assert abs(la[k] - lc[k]) <= tol, (k, la[k], l2[k], lc[k])
    # control: a resume WITHOUT the optimizer state is far outside the floor (the check above has teeth)
    m2 = _synth_model()
    d = Trainer(m2, **kw)
    load_checkpoint(m2, f, strict=True)
    run(d, range(2, 4))
    torch.cuda.synchronize()
    d_ctrl = (a.arena.momentum - d.arena.momentum).norm().item() / a.arena.momentum.norm().item()
    # (two identical runs differ by 3.5-6 % in the momentum norm after four steps -- atomics order, flipped NMS decisions -- and a
    #  single loss of a later step by up to ~5 %; the control sits at 0.3-0.4: compared with what the RESUMED run measured, not
    #  with a multiple of one noise sample)
    assert d_ctrl > 0.2 and d_ctrl > 3 * d_m, (d_ctrl, d_m, floor_m)


def test_graph_features_match_eager(first_k):
    """bonai_amd/graphs.py: backbone + neck forward and backward recorded into two hipGraphs at the trainer's third step and
    replayed from then on.  With lr = 0 every step must reproduce plain autograd's gradients -- the two eager steps before the
    capture, the step that captures, and the replays after it -- and a changed image must change the features (the replay reads
    the static input buffer, not a stale copy)."""
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    data = make_batch(2, 256, 8, device='cuda')
    want = _autograd_grads(_synth_model(), data)
    m = _synth_model()
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0, graph_features=True)
    for step in range(5):
        lv = dict(tr.train_step(data, lr=0.0)['log_vars'].items())
        torch.cuda.synchronize()
        assert tr._fgraphs is not None and tr._fgraphs.failed is None, tr._fgraphs.failed
        assert tr._fgraphs.ready == (step >= 2)
        _compare(want, m, f'graph step {step}')
    other = make_batch(2, 256, 8, step=3, device='cuda')
    want2 = _autograd_grads(_synth_model(), other)
    tr.train_step(other, lr=0.0)
    torch.cuda.synchronize()
    _compare(want2, m, 'graph replay on another batch')
    assert m.feat_provider is None                              # only set inside train_step
    with pytest.raises(Exception, match='captured for images'):
        tr.train_step(make_batch(2, 320, 8, device='cuda'), lr=0.0)


def test_loss_decreases_on_a_fixed_batch():
    """Runner parity, end to end: 40 optimisation steps of the configured recipe (SGD momentum 0.9, weight decay
    1e-4, gradient clip 35, linear warm-up; schedule_2x_bonai.py:2-10) on ONE fixed batch with the random sampler -- the model
    must fit it: every head's loss falls, the total by more than a third, nothing goes non-finite.  A sign error, a stale
    packing (weights updated in the arena but not re-packed) or a dropped gradient deposit cannot pass this."""
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer, step_lr
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    cfg = _cfg()
    torch.manual_seed(0)
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    tr = Trainer(m, lr=0.002, momentum=cfg.optimizer.momentum, weight_decay=cfg.optimizer.weight_decay,
                 max_norm=cfg.optimizer_config.grad_clip.max_norm)
    data = make_batch(2, 256, 8, device='cuda')
    logs = []
    for it in range(40):
        out = tr.train_step(data, lr=step_lr(0.002, it, 0, warmup_iters=10, warmup_ratio=0.001))
        logs.append(dict(out['log_vars'].items()))
    first = {k: sum(l[k] for l in logs[:4]) / 4 for k in logs[0]}
    last = {k: sum(l[k] for l in logs[-4:]) / 4 for k in logs[0]}
    print('fixed-batch fit, mean of first / last four steps:', {k: (round(first[k], 4), round(last[k], 4)) for k in first})
    assert all(v == v and abs(v) < 1e6 for l in logs for v in l.values())
    assert last['loss'] < 0.67 * first['loss'], (first['loss'], last['loss'])
    for k in ('loss_rpn_cls', 'loss_cls', 'loss_mask', 'loss_offset'):
        assert last[k] < first[k], (k, first[k], last[k])
    assert torch.isfinite(tr.arena.data).all()


def test_training_trajectory_vs_cpu_oracle(first_k):
    """Runner parity against the reference's recipe itself: the CPU oracle (fp32, oracle/loft_model_ref.forward_train under
    torch autograd) is trained for 5 steps with what mmcv's OptimizerHook + torch.optim.SGD do (apis/train.py:84-108:
    clip_grad_norm_(35) then SGD(momentum 0.9, weight_decay 1e-4); frozen stem / layer1, frozen BN statistics) and the HIP
    Trainer (bf16 kernels, fused clip + SGD over the flat arena) runs the same 5 steps on the same batch: the loss trajectories
    must agree step by step (the single-step bf16 tolerances of test_e2e_gpu.py, 2-5 % per loss) and the parameter UPDATES after 5 steps must
    point the same way (cosine > 0.97 for the layers checked, update norms within 10 %)."""
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    from oracle.synth_weights import synth_tensor
    steps, lr = 5, 5e-4     # (at 2e-3 the name-seeded weights overshoot -- total loss 11.6 -> 124.7 -> 72.0 -> 6.7 -> 39.5 -- and
                            #  BOTH paths follow that same trajectory within 3 %; the committed test uses a calmer rate)
    m = _synth_model()
    init = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    for n in trainable:
        sd[n] = sd[n].clone().requires_grad_(True)
    opt = torch.optim.SGD([sd[n] for n in trainable], lr=lr, momentum=0.9, weight_decay=1e-4)
    cpu = make_batch(2, 256, 8)
    ref_logs = []
    for _ in range(steps):
        opt.zero_grad()
        losses = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'])
        losses['loss'].backward()
        torch.nn.utils.clip_grad_norm_([sd[n] for n in trainable if sd[n].grad is not None], 35.0)
        opt.step()
        ref_logs.append({k: float(v.sum()) for k, v in losses.items()})
    tr = Trainer(m, lr=lr, momentum=0.9, weight_decay=1e-4, max_norm=35.0)
    data = make_batch(2, 256, 8, device='cuda')
    logs = [dict(tr.train_step(data)['log_vars'].items()) for _ in range(steps)]
    torch.cuda.synchronize()
    # per-loss tolerances of the single-step bf16 comparison (tests/test_e2e_gpu.py), widened by half for steps 2-5
    tol = dict(loss_rpn_cls=0.02, loss_rpn_bbox=0.05, loss_cls=0.03, loss_bbox=0.05, loss_mask=0.03, loss_offset=0.05, loss=0.05)
    print('loss trajectory HIP / oracle:', [(round(a['loss'], 3), round(b['loss'], 3)) for a, b in zip(logs, ref_logs)])
    for i, (a, b) in enumerate(zip(logs, ref_logs)):
        for k, t in tol.items():
            assert abs(a[k] - b[k]) <= (t if i == 0 else 1.5 * t) * max(1.0, abs(b[k])), (i, k, a[k], b[k])
    now = dict(m.named_parameters())
    report = {}
    for n in ('backbone.layer2.0.conv1.weight', 'backbone.layer4.2.conv3.weight', 'backbone.layer3.1.bn2.weight',
              'neck.fpn_convs.0.conv.weight', 'rpn_head.rpn_conv.weight', 'roi_head.bbox_head.shared_fcs.0.weight',
              'roi_head.mask_head.convs.3.conv.weight', 'roi_head.offset_head.expand_convs.2.5.weight',
              'roi_head.offset_head.fcs.1.weight', 'roi_head.offset_head.fc_offset.bias'):
        ua = (now[n].detach().cpu() - init[n]).flatten()
        ub = (sd[n].detach() - init[n]).flatten()
        cos = float(torch.dot(ua, ub) / (ua.norm() * ub.norm() + 1e-30))
        report[n] = (round(cos, 4), round(float(ua.norm() / (ub.norm() + 1e-30)), 4))
        # (the two-element fc_offset bias moves by 1e-4 in five steps; its update length follows the run-to-run spread of the loss
        #  trajectory itself -- 18.02 .. 18.34 at step 1 over the rounds' runs -- more than any tensor with thousands of entries)
        lo, hi = (0.8, 1.25) if n.endswith('fc_offset.bias') else (0.9, 1.1)
        assert cos > 0.97 and lo < float(ua.norm() / (ub.norm() + 1e-30)) < hi, (n, report[n])
    print('update after 5 steps, HIP vs CPU oracle (cosine, norm ratio):', report)


def test_graph_replay_follows_the_optimizer_and_survives_reloaded_frozen_weights(first_k):
    """(a) with lr > 0 and momentum the replayed graphs must read the weights the SGD kernel just wrote (operand
    packings re-made every step into the buffers whose addresses the graphs hold) -- five optimisation steps under
    graph_features=True against the same five steps on eager launches: same loss trajectory, same parameter updates; (b) a
    FROZEN weight replaced after capture (load_state_dict mid-run: its cached packing, whose address is baked into the graphs,
    is stale) is noticed before the next replay -- the graphs are dropped with a warning and recaptured, and the step's
    gradients equal plain autograd's on the new weights; (c) a model that has run can be pickled (pack cache entries hold weak
    references)."""
    import copy
    import pickle
    import warnings
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    data = make_batch(2, 256, 8, device='cuda')
    logs, finals = {}, {}
    for mode in ('eager', 'graph'):
        m = _synth_model()
        init = {n: p.detach().clone() for n, p in m.named_parameters()}
        tr = Trainer(m, lr=5e-4, momentum=0.9, weight_decay=1e-4, max_norm=35.0, graph_features=(mode == 'graph'))
        logs[mode] = [dict(tr.train_step(data)['log_vars'].items()) for _ in range(5)]
        torch.cuda.synchronize()
        if mode == 'graph':
            assert tr._fgraphs.ready and tr._fgraphs.failed is None and tr._fgraphs.pack_deps, tr._fgraphs.failed
        finals[mode] = {n: (p.detach() - init[n]).flatten() for n, p in m.named_parameters() if p.requires_grad}
    print('loss trajectory eager / graph:', [(round(a['loss'], 3), round(b['loss'], 3)) for a, b in zip(logs['eager'], logs['graph'])])
    # per-loss tolerances of the trajectory test against the CPU oracle (two correct bf16 runs of this model drift apart by a
    # few percent within a handful of steps: split-K atomics reorder sums, the name-seeded weights amplify); a replay on stale
    # operands would repeat step 2's losses and leave the trajectory at once
    tol = dict(loss_rpn_cls=0.03, loss_rpn_bbox=0.075, loss_cls=0.045, loss_bbox=0.075, loss_mask=0.045, loss_offset=0.075, loss=0.075, acc=0.05)
    for i, (a, b) in enumerate(zip(logs['eager'], logs['graph'])):
        for k in a:
            assert abs(a[k] - b[k]) <= tol.get(k, 0.075) * max(1.0, abs(a[k])), (i, k, a[k], b[k])
    tail = [logs['graph'][i]['loss'] for i in (2, 3, 4)]                      # the replays see new weights: the losses keep moving
    assert max(tail) - min(tail) > 0.005 * tail[0], tail                      # (one difference of two steps can sit near a turning point)
    for n in ('backbone.layer2.0.conv1.weight', 'backbone.layer4.2.conv3.weight', 'backbone.layer3.1.bn2.weight',
              'neck.fpn_convs.0.conv.weight', 'neck.lateral_convs.2.conv.weight'):
        ua, ub = finals['eager'][n], finals['graph'][n]
        cos = float(torch.dot(ua, ub) / (ua.norm() * ub.norm() + 1e-30))
        assert cos > 0.98 and 0.93 < float(ua.norm() / (ub.norm() + 1e-30)) < 1.07, (n, cos)
    # (b) lr = 0 from here on; replace a frozen stem / layer1 weight in place of the captured one
    m = _synth_model()
    tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0, graph_features=True)
    for _ in range(3):
        tr.train_step(data, lr=0.0)
    assert tr._fgraphs.ready
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad and n.startswith('backbone.layer1') and n.endswith('conv2.weight')]
    assert frozen
    sd[frozen[0]] = sd[frozen[0]] * 1.5
    m.load_state_dict(sd)
    ref = _synth_model()
    ref.load_state_dict(sd)
    want = _autograd_grads(ref, data)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        tr.train_step(data, lr=0.0)
        torch.cuda.synchronize()
    assert any('dropped and recaptured' in str(x.message) for x in w), [str(x.message) for x in w]
    assert tr._fgraphs.ready and tr._fgraphs.stale is not None
    _compare(want, m, 'after the frozen weight changed under a live graph')
    tr.train_step(data, lr=0.0)                                    # and the recaptured graphs replay
    torch.cuda.synchronize()
    _compare(want, m, 'replay of the recaptured graphs')
    # (c)
    blob = pickle.dumps(m.backbone.conv1)
    assert pickle.loads(blob).weight.shape == m.backbone.conv1.weight.shape
    copy.deepcopy(m.backbone.layer1)


@pytest.mark.parametrize('cfg_name', ['loft_foa_r50_fpn_mdconv_c3-c5_2x_bonai.py', 'loft_foa_hrnetv2p_w32_2x_bonai.py'])
def test_graph_replay_follows_the_optimizer_side_configs(cfg_name, first_k):
    """the lr > 0 replay check on the other two backbones -- DCNv2 (config 4: the deformable sampler's workspace and
    offset convs inside the captured section) and HRNet-W32 + HRFPN (config 5, bf16 here: branch streams forked inside the graph).
    Four optimisation steps under graph_features=True against the same steps on eager launches: same loss trajectory, and the
    replays see the weights the SGD kernel wrote (the trajectory moves)."""
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.synth import make_batch
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', cfg_name))
    data = make_batch(2, 256, 8, device='cuda')
    logs = {}
    for mode in ('eager', 'graph'):
        m = _synth_model(cfg)
        tr = Trainer(m, lr=5e-4, momentum=0.9, weight_decay=1e-4, max_norm=35.0, graph_features=(mode == 'graph'))
        logs[mode] = [dict(tr.train_step(data)['log_vars'].items()) for _ in range(4)]
        torch.cuda.synchronize()
        if mode == 'graph':
            assert tr._fgraphs.ready and tr._fgraphs.failed is None, tr._fgraphs.failed
    print(cfg_name, 'loss trajectory eager / graph:', [(round(a['loss'], 3), round(b['loss'], 3)) for a, b in zip(logs['eager'], logs['graph'])])
    for i, (a, b) in enumerate(zip(logs['eager'], logs['graph'])):
        assert all(v == v and abs(v) < 1e6 for v in b.values())
        assert abs(a['loss'] - b['loss']) <= 0.08 * max(1.0, abs(a['loss'])), (i, a['loss'], b['loss'])
    assert abs(logs['graph'][3]['loss'] - logs['graph'][2]['loss']) > 1e-3 * logs['graph'][2]['loss']


@pytest.mark.parametrize('lateral_first', [False, True])
def test_gradient_join_is_order_safe_and_matches_the_autograd_sum(lateral_first):
    """nn.JOIN: a stage output x with two consumers -- a strided bottleneck (conv shortcut) and a plain 1x1 conv (the FPN lateral).
    With the join open, the conv's data gradient is deposited and rides in the block's first data-gradient launch when autograd
    runs the conv first (it was created later: lateral_first=False, the model's order); when the conv was created FIRST its backward
    runs after the block's, finds no open entry and returns its gradient to autograd as before.  Either way dL/dx equals the plain
    autograd sum (bf16 rounding of one add apart) and every parameter gradient is unchanged."""
    from bonai_amd import kernels as K
    from bonai_amd import nn as F2
    from bonai_amd.loft.backbone import Bottleneck, ConvW
    torch.manual_seed(5)
    blk = Bottleneck(256, 128, stride=2, downsample=True).cuda()
    lat = ConvW(256, 256, 1, bias=True).cuda()
    for p in list(blk.parameters()) + list(lat.parameters()):
        if p.dim() > 1:
            torch.nn.init.normal_(p, 0, 0.05)
    xin = torch.randn(2, 256, 64, 64, device='cuda')

    def run(join):
        for p in list(blk.parameters()) + list(lat.parameters()):
            p.grad = None
        x0 = xin.clone().requires_grad_(True)
        x = torch.relu(x0).to(K.L.act16()).contiguous(memory_format=torch.channels_last)
        F2.JOIN = {} if join else None
        try:
            if lateral_first:
                a = F2.conv2d(x, lat.weight, lat.bias)
                b = blk(x)
            else:
                b = blk(x)
                a = F2.conv2d(x, lat.weight, lat.bias)
            if join:
                assert x.data_ptr() in F2.JOIN and F2.JOIN[x.data_ptr()] is None
            (a.float().square().mean() + b.float().square().mean()).backward()
            if join:
                assert x.data_ptr() not in F2.JOIN            # the block popped its entry
        finally:
            F2.JOIN = None
        # (dL/dx0, behind the producer's ReLU backward: the joined path hands the producer a gradient that is already masked
        #  with x > 0 -- the block's epilogue -- the plain path an unmasked sum; the mask is idempotent)
        return x0.grad.float().clone(), [p.grad.clone() for p in list(blk.parameters()) + list(lat.parameters())]

    gx0, pg0 = run(False)
    gx1, pg1 = run(True)
    assert (gx1 - gx0).norm().item() <= 6e-3 * gx0.norm().item()      # one bf16 rounding less on the joined path
    for a, b in zip(pg1, pg0):
        assert (a - b).norm().item() <= 1e-2 * b.norm().item() + 1e-12
