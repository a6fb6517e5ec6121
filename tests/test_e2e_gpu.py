"""GPU parity, end to end: the HIP LOFT training step vs (a) the fixture produced by the REFERENCE's own
python (tests/golden/e2e_256.npz) and (b) the CPU oracle, on the same seeded tile, name-keyed weights and the
injected 'first-k' sampling rule.

Tolerances: the north-star tolerance of 1e-3 applies to fp32 arithmetic; this path computes its contractions
with bf16 operands (fp32 accumulation) through ~60 stacked layers, so end-to-end quantities are compared at
bf16-accumulated tolerances, stated per quantity below.  fp32-exact pieces (assignment indices, NMS keep
lists, coders, RoIAlign, targets) are pinned at 1e-5 / bit-exact in the per-kernel tests.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


# Tolerances of the bf16 end-to-end comparison against the reference-made fixture (tests/golden/e2e_256.npz): 1.5 x what the shipped
# kernels measure (the measured values are printed by the test and recorded in
# profiles/round5_probes/bf16_e2e_measured.txt), so that a regression cannot hide inside a generous bound.
# Measured (three runs, identical to the digit): features 0.0467 of the map's mean magnitude; losses rpn_cls 5.3e-4, rpn_bbox 1e-4,
# cls 3.7e-3, bbox 1.3e-3, mask 6.2e-3, offset 1.7e-3, total 2.3e-3; gradient norms 5.5e-3; leading entries inside the 0.5 rms floor.
# (Rounds 1-4 allowed 12 % / 2-5 % / 8 % / 12 %.)
TOL_BF16 = dict(feat=0.07, loss=dict(loss_rpn_cls=2e-3, loss_rpn_bbox=2e-3, loss_cls=6e-3, loss_bbox=3e-3, loss_mask=1e-2, loss_offset=3e-3,
                                     loss=4e-3), gradnorm=9e-3, gradhead=0.03)


def _build(sample_first=True):
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first' if sample_first else 'random'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    return m.cuda().train()


# Round 6: the default path runs the last 1x1 of bottleneck k and the first 1x1 of bottleneck k+1 as one launch (loft_bneck_pair_bf16).
# At kernel level it equals the separate launches except for ~1 element in 5e5 by one bf16 unit (the residual is added on the matrix
# pipe; tests/test_bneck_pair_gpu.py) -- but this random-weight detector selects proposals by score order and IoU thresholds, and ONE
# flipped selection moves the losses by 1e-3 and single gradient norms by percents: with the fusion the distances to the reference
# are features 0.0467, losses <= 5.7e-3 (inside the bounds above), gradient norms 2.13e-2 (0.55e-2 without;
# tools/probes/e2e_measured.py, identical run to run).  The test therefore runs BOTH ways: the separate launches at the bounds above
# (a regression of any kernel they share cannot hide), the fused default at 1.5 x ITS measured gradient-norm distance.
GRADNORM_PAIR = 3.2e-2


@pytest.mark.parametrize('pair_fusion', [False, True])
def test_e2e_losses_features_grads_vs_reference_fixture(pair_fusion):
    from bonai_amd.debug import DBG
    with DBG.override(no_pair_fusion=not pair_fusion):
        _e2e_vs_fixture(dict(TOL_BF16, gradnorm=GRADNORM_PAIR) if pair_fusion else TOL_BF16)


def _e2e_vs_fixture(TOL_BF16):
    from bonai_amd.synth import make_batch
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    m = _build()
    data = make_batch(batch, size, num_gt, device='cuda')
    feats = m.extract_feat(data['img'])
    meas = dict(feat=0.0, loss={}, gradnorm=0.0, gradhead=0.0)
    for i, f in enumerate(feats):
        want = torch.from_numpy(gd[f'feat_{i}_crop'])
        got = f[:, :8, :6, :6].float().cpu()
        scale = float(gd[f'feat_{i}_absmean'])
        meas['feat'] = max(meas['feat'], (got - want).abs().max().item() / scale)
        # bf16 operands through up to 53 convs, worst entry of a 6 x 6 x 8 crop relative to the map's mean magnitude: TOL_FEAT
        assert (got - want).abs().max().item() < TOL_BF16['feat'] * scale, (i, (got - want).abs().max().item(), scale)
    out = m.train_step(data)
    lv = dict(out['log_vars'].items())
    for k, t in TOL_BF16['loss'].items():
        want = float(gd['log_' + k])
        meas['loss'][k] = abs(lv[k] - want) / max(1.0, abs(want))
        assert abs(lv[k] - want) <= t * max(1.0, abs(want)), (k, lv[k], want)
    assert abs(lv['acc'] - float(gd['log_acc'])) <= 3.0
    out['loss'].backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for k in gd.files:
        if k.startswith('gradnorm_'):
            n = k[len('gradnorm_'):]
            want = float(gd[k])
            got = float(grads[n].norm()) if n in grads else 0.0
            meas['gradnorm'] = max(meas['gradnorm'], abs(got - want) / max(1e-2, want))
            assert abs(got - want) <= TOL_BF16['gradnorm'] * max(1e-2, want), (n, got, want)
            # element level (first 16 gradient entries of the reference run): a sign / permutation error inside a composite
            # autograd node cannot hide behind a matching norm.  bf16 chain: a fraction of the head's own norm + a floor of the
            # tensor's RMS entry.
            wh = torch.from_numpy(gd['gradhead_' + n])
            gh = grads[n].float().reshape(-1)[:16].cpu()
            rms = want / grads[n].numel() ** 0.5
            meas['gradhead'] = max(meas['gradhead'], max(0.0, (gh - wh).norm().item() - 0.5 * rms) / max(wh.norm().item(), 1e-12))
            assert (gh - wh).norm().item() <= TOL_BF16['gradhead'] * wh.norm().item() + 0.5 * rms, (n, gh, wh)
    print('bf16 e2e vs the reference fixture, measured / allowed:', {k: (round(v, 5) if not isinstance(v, dict) else {a: round(b, 5) for a, b in v.items()})
                                                                   for k, v in meas.items()}, TOL_BF16)


def test_e2e_vs_oracle_proposals_and_targets():
    """Same tile through the CPU oracle: proposals (after NMS) must agree as sets up to bf16 score jitter."""
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M, ops_ref as R
    from oracle.synth_weights import synth_tensor
    m = _build()
    data = make_batch(2, 256, 10, device='cuda')
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    with torch.no_grad():
        x = m.extract_feat(data['img'])
        fused = m.rpn_head.forward_fused(x)
        props, counts = m.rpn_head.get_bboxes_fused(fused, data['img_metas'], m.train_cfg.rpn_proposal)
        cpu = make_batch(2, 256, 10)
        feats = M.fpn(sd, M.backbone(sd, cpu['img']))
        cls, reg = M.rpn_forward(sd, feats)
        ref = M.rpn_proposals(cls, reg, (256, 256, 3))
    for i in range(2):
        n = int(counts[i])
        assert abs(n - ref[i].shape[0]) <= 0.02 * ref[i].shape[0] + 2
        got = props[i, :n].cpu()
        assert (got[:-1, 4] >= got[1:, 4]).all()
        # every one of the reference's top-100 proposals has a near-identical box among ours
        rb = ref[i][:150, :4]
        rb = rb[((rb[:, 2] - rb[:, 0]) * (rb[:, 3] - rb[:, 1])) > 4.0]     # zero-area (clamped) boxes have IoU 0 with anything
        iou = R.bbox_overlaps(rb, got[:, :4])
        assert (iou.max(dim=1)[0] > 0.85).float().mean().item() > 0.9


def test_e2e_fp32_parity_mode_vs_reference_fixture(f32_contract):
    """North-star tolerance (1e-3), forward AND backward: the same fixture as above with the fp32 parity mode (fp32 activations)
    -- features, all seven losses and the gradient of every parameter at 1e-3, for the mode's default contraction (SPLIT6: three
    bf16 per fp32 operand, six MFMA terms) and for the exact fp32 MFMA.  SPLIT3 (two bf16 per operand: 16 mantissa bits) meets
    1e-3 on features and losses; on the gradients of this random-weight network its 2e-6 per layer grows to 2e-3 (median 3e-4:
    tools/probes/f32_grad_errors.py) and single entries of the most cancellation-prone gradient (layer3.0.downsample, 6e-3 even in
    exact fp32) to 5e-2: its backward is a sanity check here (norms 5e-3, entries 1e-1), not a parity claim."""
    from bonai_amd.synth import make_batch
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    m = _build()
    m.backbone.compute_dtype = torch.float32
    data = make_batch(batch, size, num_gt, device='cuda')
    with torch.no_grad():
        feats = m.extract_feat(data['img'])
        for i, f in enumerate(feats):
            assert f.dtype == torch.float32
            want = torch.from_numpy(gd[f'feat_{i}_crop'])
            got = f[:, :8, :6, :6].cpu()
            scale = float(gd[f'feat_{i}_absmean'])
            assert (got - want).abs().max().item() < 1e-3 * scale, (i, (got - want).abs().max().item(), scale)
    out = m.train_step(data)
    lv = dict(out['log_vars'].items())
    for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'loss_mask', 'loss_offset', 'loss'):
        want = float(gd['log_' + k])
        assert abs(lv[k] - want) <= 1e-3 * max(1.0, abs(want)), (k, lv[k], want)
    assert abs(lv['acc'] - float(gd['log_acc'])) <= 0.2     # one of 1024 sampled RoIs flipping is 0.1
    # BACKWARD in the parity mode (fp32 dgrad on loft_conv_tap_f32, fp32 weight gradient / glue adjoints of parity_f32.hip): the
    # gradient of EVERY trainable parameter against the reference's own autograd -- norm at 1e-3, leading entries at 1e-2 of the
    # gradient's scale.  This pins the composition of the autograd nodes (residual-block node, RoIAlign / FPN adjoints, narrow
    # heads, grouped FOA launches), not just the kernels one by one.
    out['loss'].backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    names = [k[len('allnorm_'):] for k in gd.files if k.startswith('allnorm_')]
    assert sorted(names) == sorted(n for n, g in grads.items() if g is not None) and len(names) > 200
    worst = worst1 = worst2 = (0.0, None)
    for n in names:
        g = grads[n].float()
        wn = float(gd['allnorm_' + n])
        gn = float(g.norm())
        rms = wn / max(g.numel(), 1) ** 0.5
        wh = torch.from_numpy(gd['allhead_' + n])
        gh = g.reshape(-1)[:wh.numel()].cpu()
        e1 = abs(gn - wn) / max(wn, 1e-12)
        e2 = float((gh - wh).abs().max()) / max(float(wh.abs().max()), rms, 1e-12)
        worst = max(worst, (e1, n), (e2, n))
        worst1, worst2 = max(worst1, (e1, n)), max(worst2, (e2, n))
    print(f'fp32 parity backward ({f32_contract}): worst norm error {worst1}, worst leading entry {worst2}')
    assert worst1[0] <= (5e-3 if f32_contract == 'split3' else 1e-3), ('norm', worst1)
    # single entries: 1e-2 of max(largest listed entry, rms) -- the entries are sums of ~1e5 signed fp32 products whose order
    # differs from the reference's CPU kernels, on top of the few mask-target pixels that sit on the 0.5 edge.  The binary16
    # planes carry 22-23 significant bits per operand (measured 1.6e-2 with or without the lo x lo product: the operand
    # representation, not the dropped term, is what separates them from the 24-bit contractions' 5-6e-3): 2e-2.
    assert worst2[0] <= dict(split3=1e-1, planes_f16=2e-2, planes_f16x4=2e-2).get(f32_contract, 1e-2), ('head', worst2)


def test_sparse_rpn_backward_matches_dense_autograd():
    """The sampled-anchor-only backward of the RPN head (bonai_amd.nn._SparseRPNFn) against the plain dense autograd path
    on the same step: identical losses, gradients equal up to bf16 accumulation order."""
    from bonai_amd.synth import make_batch
    data = make_batch(2, 256, 10, device='cuda')
    grads, losses = [], []
    for sparse in (True, False):
        m = _build()
        m.rpn_head.sparse_backward = sparse
        out = m.train_step(data)
        out['loss'].backward()
        losses.append({k: float(v) for k, v in out['log_vars'].items()})
        grads.append({n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None})
    for k in losses[0]:
        assert abs(losses[0][k] - losses[1][k]) <= 1e-5 * max(1.0, abs(losses[1][k])), k
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        tol = 0.03 if n.startswith('rpn_head') else 0.05     # (backbone/neck: plus the bf16 atomics of the scattered dx)
        assert (a - b).norm().item() <= tol * max(b.norm().item(), 1e-6), (n, (a - b).norm().item(), b.norm().item())


def test_reference_format_checkpoints_through_the_hip_model(tmp_path):
    """(i) a checkpoint in the layout mmcv's CheckpointHook writes ({'meta', 'state_dict', 'optimizer'}, keys with the
    'module.' prefix of MMDistributedDataParallel; apis/train.py:139-142, tools/publish_model.py:16-30) and (ii) a torchvision-keyed
    backbone file handed over as ``pretrained`` (bonai_loft_foa_r50_fpn_basic.py:4, resnet.py:591-600) are loaded through
    bonai_amd.checkpoint.load_checkpoint into the HIP model, which must then reproduce the reference-generated fixture."""
    from collections import OrderedDict
    from bonai_amd.checkpoint import load_checkpoint
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle.synth_weights import synth_tensor
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    # (ii) torchvision layout: backbone keys without prefix, plus the classifier the detector does not have
    probe = build_detector(dict(cfg.model, pretrained=None), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    names = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
    tv = OrderedDict((k[len('backbone.'):], synth_tensor(k, s)) for k, s in names.items() if k.startswith('backbone.'))
    tv['fc.weight'], tv['fc.bias'] = torch.zeros(1000, 2048), torch.zeros(1000)
    tv_path = str(tmp_path / 'resnet50-synth.pth')
    torch.save(tv, tv_path)
    torch.manual_seed(123)                                      # everything else starts from some OTHER random init
    RandomSampler.choice_mode = 'first'
    m = build_detector(dict(cfg.model, pretrained=tv_path), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    for k, v in m.backbone.state_dict().items():
        assert torch.equal(v.cpu(), tv[k]), k
    # (i) reference training checkpoint of the whole detector, DDP-prefixed, with optimizer state and meta
    ck = dict(meta=dict(mmdet_version='2.3.0', config='(text)', CLASSES=('building',), epoch=3, iter=1234),
              state_dict=OrderedDict(('module.' + k, synth_tensor(k, s)) for k, s in names.items()),
              optimizer=dict(state={}, param_groups=[dict(lr=0.005, momentum=0.9, weight_decay=1e-4, params=list(range(len(names))))]))
    ck_path = str(tmp_path / 'epoch_3.pth')
    torch.save(ck, ck_path)
    out = load_checkpoint(m, ck_path, strict=True)
    assert out['meta']['iter'] == 1234
    m = m.cuda().train()
    data = make_batch(batch, size, num_gt, device='cuda')
    lv = dict(m.train_step(data)['log_vars'].items())
    tol = dict(loss_rpn_cls=0.02, loss_rpn_bbox=0.05, loss_cls=0.03, loss_bbox=0.05, loss_mask=0.03, loss_offset=0.05, loss=0.05)
    for k, t in tol.items():
        want = float(gd['log_' + k])
        assert abs(lv[k] - want) <= t * max(1.0, abs(want)), (k, lv[k], want)
    RandomSampler.choice_mode = 'random'


@pytest.mark.parametrize('size,batch', [(384, 2), (320, 3)])
def test_odd_map_sizes_16bit_backward_agrees_with_fp32_parity_backward(size, batch):
    """Feature maps of 96 / 48 / 24 / 12 (80 / 40 / 20 / 10) pixels: none of the bench's power-of-two widths, so the decode-free
    weight-gradient modes take other branches (same-size taps need OW >= 32 in the ring kernel, >= 64 in the stream kernel; rows of
    a K-step then wrap image rows differently) and the pixel-major RoI blocks are ragged.  The 16-bit training path must agree
    with the fp32 parity path -- different kernels throughout -- at 16-bit tolerances: every loss, every gradient norm."""
    from bonai_amd.synth import make_batch
    m = _build()
    data = make_batch(batch, size, 9, device='cuda')
    res = {}
    for mode, dt in (('f32', torch.float32), ('b16', None)):
        m.backbone.compute_dtype = dt
        m.zero_grad(set_to_none=True)
        out = m.train_step(data)
        out['loss'].backward()
        res[mode] = (dict(out['log_vars'].items()), {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None})
    lf, gf = res['f32']
    lb, gb = res['b16']
    for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'loss_mask', 'loss_offset'):
        assert abs(lb[k] - lf[k]) <= 0.05 * max(1.0, abs(lf[k])), (k, lb[k], lf[k])
    assert set(gf) == set(gb) and len(gf) > 200
    bad = []
    for n, g32 in gf.items():
        n32, n16 = float(g32.norm()), float(gb[n].norm())
        # (a one-element gradient -- the mask logits' bias -- is a sum of ~1e5 signed terms that largely cancel: 14 % measured)
        if abs(n16 - n32) > (0.25 if g32.numel() <= 8 else 0.10) * max(n32, 1e-2):
            bad.append((n, n16, n32))
    assert not bad, bad[:8]


def _rel_l2(got, want):
    return float((got.float().cpu() - want).norm() / want.norm())


def test_e2e_bf16_kernels_vs_bf16_points_oracle():
    """the TIMED bf16 kernels bounded at model level AND per stage.

    The CPU oracle runs the same tile in its 16-bit-points mode (oracle/loft_model_ref.numerics: bf16 roundings exactly where
    this path holds bf16 data, fp32 accumulation; pinned by tests/golden/e2e_256_bf16.npz).  What a bound can be: two bf16
    pipelines that sum in different orders do NOT stay element-identical -- a value that lands on the other side of a rounding
    boundary (1 ulp = 2^-8) perturbs ~2300 products of the next layer, which flips more roundings; within a few layers the two
    rounding-error fields are independent.  Measured: the SAME oracle code on two x86 hosts differs by 6.6e-3 relative L2 on the
    FPN maps, the oracle's bf16-points mode sits 8.7e-3 from its own fp32 mode, this path 8.2e-3 from the bf16-points oracle.
    So (a) end to end, the HIP path must be no farther from the bf16-points oracle than bf16 rounding noise itself puts two
    correct implementations apart: <= 1.5 x d(oracle16, oracle32) per FPN map, seven losses within 1e-2; and (b) PER STAGE, fed
    with the oracle's own (bf16-valued) stage input so that nothing accumulates across stages: each stage's output within
    max(2e-3, 1.3 x that stage's own d(oracle16, oracle32)) -- a wrong rounding point, a missing ReLU or a mis-folded BN in any
    stage shows up as a multiple of that."""
    from bonai_amd import kernels as K
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M, ops_ref as R
    from oracle.synth_weights import synth_tensor
    gd = np.load(os.path.join(GOLD, 'e2e_256_bf16.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    m = _build()
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    cpu = make_batch(batch, size, num_gt)
    b16 = torch.bfloat16
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    dev = lambda t: t.to('cuda', b16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        with M.numerics(b16):
            ol, ex = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'],
                                     return_extras=True)
        l32, e32 = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'],
                                   return_extras=True)
    noise = []
    for i in range(5):     # the live oracle IS the committed fixture -- up to THIS host's conv summation order
        want = torch.from_numpy(gd[f'feat_{i}_sub'])
        noise.append(float((ex['feats'][i][:, ::8, ::2, ::2] - want).norm() / want.norm()))
        assert noise[-1] < 1.5e-2, noise
    # ---- (b) per stage, on the oracle's inputs
    rows = []

    def stage(name, hip_out, o16, o32):
        d, d0 = rel(hip_out, o16), float((o16 - o32).norm() / o32.norm())
        rows.append((name, d, d0))
    with torch.no_grad():
        x0_16 = M.stem(sd, cpu['img'].to(b16).float())          # (16-bit mode rounds the image itself; same input for fp32)
        with M.numerics(b16):
            xs16 = [M.stem(sd, cpu['img'])]
            for li in range(4):
                xs16.append(M.res_stage(sd, xs16[-1], li))
            p16 = M.fpn(sd, xs16[1:])
            rc16, rr16 = M.rpn_forward(sd, p16)
        # fp32 arithmetic of every stage on the SAME bf16-valued input
        xs32 = [x0_16] + [M.res_stage(sd, xs16[li], li) for li in range(4)]
        p32 = M.fpn(sd, xs16[1:])
        rc32, rr32 = M.rpn_forward(sd, p16)
        bb = m.backbone
        scale, shift = bb.bn1.fold()
        hx = K.maxpool3x3s2(K.stem7x7_mfma(cpu['img'].cuda(), bb.conv1.weight, scale, shift))
        stage('stem+pool', hx, xs16[0], xs32[0])
        for li in range(4):
            hy = getattr(bb, f'layer{li + 1}')(dev(xs16[li]))
            stage(f'layer{li + 1}', hy, xs16[li + 1], xs32[li + 1])
        hp = m.neck(tuple(dev(c) for c in xs16[1:]))
        for i in range(5):
            stage(f'fpn P{i + 2}', hp[i], p16[i], p32[i])
        hf = m.rpn_head.forward_fused([dev(p) for p in p16])
        A = m.rpn_head.num_anchors
        for i in range(5):
            stage(f'rpn head P{i + 2}', hf[i][:, :5 * A], torch.cat([rc16[i], rr16[i]], 1), torch.cat([rc32[i], rr32[i]], 1))
        rh = m.roi_head
        p4 = p16[:4]
        xr = R.roi_extract(p4, ex['rois'], 7).to(b16).float()
        xm = R.roi_extract(p4, ex['pos_rois'], 14).to(b16).float()
        xo = R.roi_extract(p4, ex['pos_rois'], 7).to(b16).float()
        with M.numerics(b16):
            cs16, bp16 = M.bbox_head(sd, xr)
            mp16, op16 = M.mask_head(sd, xm), M.foa_head(sd, xo)
        cs32, bp32 = M.bbox_head(sd, xr)
        mp32, op32 = M.mask_head(sd, xm), M.foa_head(sd, xo)
        hc, hb = rh.bbox_head(dev(xr))
        stage('bbox head cls', hc, cs16, cs32)
        stage('bbox head reg', hb, bp16, bp32)
        stage('mask head', rh.mask_head(dev(xm))[:, :1], mp16, mp32)
        x4 = torch.cat([torch.rot90(xo, k, (2, 3)) for k in range(4)], 0)
        stage('FOA head', rh.offset_head.forward_rotated(dev(x4)), op16, op32)
    print('\nstage                 HIP vs oracle16   oracle16 vs oracle32 (same input)')
    for name, d, d0 in rows:
        print(f'{name:22s}{d:12.2e}{d0:18.2e}')
    # measured (MI355X, round 3; profiles/round3_probes/bf16_stage_drift.txt): stem 1.9e-5, FPN 1.2e-4, RPN head <= 4.5e-5,
    # bbox head 2e-4, layer1 3.7e-4 -- one to three layers deep, where nothing can amplify, the roundings ARE the oracle's;
    # mask head 1.5e-3, layer2-4 2.5-4.4e-3, FOA 3.8e-3 against 4.2-5.7e-3 of rounding noise in the same stages
    shallow = ('stem', 'layer1', 'fpn', 'rpn head', 'bbox head')
    for name, d, d0 in rows:
        assert d <= (1e-3 if name.startswith(shallow) else max(2e-3, 1.3 * d0)), (name, d, d0)
    # ---- (a) end to end
    data = make_batch(batch, size, num_gt, device='cuda')
    report = {}
    with torch.no_grad():
        feats = m.extract_feat(data['img'])
        for i, f in enumerate(feats):
            report[f'feat_{i}'] = (rel(f, ex['feats'][i]), float((ex['feats'][i] - e32['feats'][i]).norm() / e32['feats'][i].norm()))
    print('FPN maps end to end, relative L2 (HIP vs oracle16, oracle16 vs oracle32):', {k: (f'{a:.2e}', f'{b:.2e}') for k, (a, b) in report.items()},
          '| oracle on this host vs the fixture\'s host:', [f'{v:.2e}' for v in noise])
    for k, (a, b) in report.items():
        assert a <= 1.5 * b, (k, a, b)
    lv = dict(m.train_step(data)['log_vars'].items())
    for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'loss_mask', 'loss_offset', 'loss'):
        want = float(ol[k].sum())
        assert abs(lv[k] - want) <= 1e-2 * max(1.0, abs(want)), (k, lv[k], want)


@pytest.mark.parametrize('batch', [1, 2])
def test_fullsize_fp32_parity_mode_vs_cpu_oracle(batch):
    """(batch 2 = BASELINE configs[0]'s workload, 2 x 1024 x 1024: a cross-image indexing error at full size cannot hide behind a
    single image -- VERDICT round 5 item 7; it runs the mode's default contraction, batch 1 runs all three.)
    VERDICT round 4, item 4: BASELINE configs[1]'s tile size against the ORACLE, not against this path's other mode.  One
    1024 x 1024 image with 80 ground-truth boxes goes through the fp32 parity mode -- under its default contraction (binary16
    operand planes on the stream kernels), the bfloat16 planes and the exact fp32 MFMA -- and ONCE through
    oracle.loft_model_ref.forward_train + torch autograd on the host (fp32; the restatement pinned to the reference by
    tests/golden/e2e_256.npz): the five FPN maps (a 24 x 24 x 16-channel crop at 1e-3 of the map's mean magnitude, the whole map
    at 1e-4 relative L2), the seven losses at 1e-3 and the gradient NORM of every trainable parameter (> 200) at 1e-3 -- the
    256 px fixture test's bounds at four times the map size, 3000 proposals and 1024 sampled RoIs.  Leading gradient ENTRIES
    are held to 3e-2 of the gradient's scale: at this size a weight-gradient entry is a sum of 10^4..10^6 signed fp32 products
    and the exact-fp32 kernels themselves sit 1e-2 from the host's summation order (printed per contraction)."""
    from bonai_amd import kernels as K
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    from oracle.synth_weights import synth_tensor
    m = _build()
    m.backbone.compute_dtype = torch.float32
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if k in trainable:
            v.requires_grad_(True)
    cpu = make_batch(batch, 1024, 80)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 8))       # (hundreds of threads oversubscribe the oracle's convolutions)
    try:
        ol, ex = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'],
                                 return_extras=True)
        ol['loss'].backward()
    finally:
        torch.set_num_threads(nthr)
    ol = {k: float(v.detach().sum()) for k, v in ol.items()}
    data = make_batch(batch, 1024, 80, device='cuda')
    if batch > 1:       # the images (and their boxes) differ: what the cross-image check needs
        assert not torch.equal(cpu['img'][0], cpu['img'][1]) and not torch.equal(cpu['gt_bboxes'][0], cpu['gt_bboxes'][1])
    names = [n for n in sorted(trainable) if sd[n].grad is not None]
    assert len(names) > 200
    prev = K.F32_CONTRACT
    report = {}
    try:
        modes = (('planes_f16', K.F32_PLANES_F16), ('planes_bf16', K.F32_PLANES_BF16), ('exact', K.F32_EXACT))
        for mode, code in (modes if batch == 1 else modes[:1]):
            K.F32_CONTRACT = code
            with torch.no_grad():
                feats = m.extract_feat(data['img'])
            frel = []
            for i, (f, w) in enumerate(zip(feats, ex['feats'])):
                w = w.detach()
                assert f.dtype == torch.float32 and tuple(f.shape) == tuple(w.shape)
                got = f.cpu()
                scale = float(w.abs().mean())
                c = min(24, w.shape[2])
                assert (got[:, :16, :c, :c] - w[:, :16, :c, :c]).abs().max().item() < 1e-3 * scale, (mode, i, scale)
                frel.append(float((got - w).norm() / w.norm()))
                assert frel[-1] < 1e-4, (mode, i, frel[-1])
            m.zero_grad(set_to_none=True)
            out = m.train_step(data)
            lv = dict(out['log_vars'].items())
            lerr = 0.0
            for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'loss_mask', 'loss_offset', 'loss'):
                lerr = max(lerr, abs(lv[k] - ol[k]) / max(1.0, abs(ol[k])))
                assert abs(lv[k] - ol[k]) <= 1e-3 * max(1.0, abs(ol[k])), (mode, k, lv[k], ol[k])
            out['loss'].backward()
            grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
            assert set(names) <= set(grads)
            w1, w2 = (0.0, None), (0.0, None)
            for n in names:
                g, w = grads[n].float().cpu(), sd[n].grad
                wn, gn = float(w.norm()), float(g.norm())
                rms = wn / max(w.numel(), 1) ** 0.5
                wh, gh = w.reshape(-1)[:16], g.reshape(-1)[:16]
                w1 = max(w1, (abs(gn - wn) / max(wn, 1e-12), n))
                w2 = max(w2, (float((gh - wh).abs().max()) / max(float(wh.abs().max()), rms, 1e-12), n))
            report[mode] = (max(frel), lerr, w1, w2)
            print(f'{batch} x 1024^2 fp32 parity mode ({mode}) vs the CPU oracle: FPN maps rel L2 <= {max(frel):.1e}, losses <= {lerr:.1e}, '
                  f'{len(names)} gradients: worst norm error {w1[0]:.2e} ({w1[1]}), worst leading entry {w2[0]:.2e} ({w2[1]})')
    finally:
        K.F32_CONTRACT = prev
    for mode, (_, _, w1, w2) in report.items():
        assert w1[0] <= 1e-3, (mode, 'norm', w1)
        assert w2[0] <= 3e-2, (mode, 'head', w2)


def test_bf16_vs_fp32_parity_mode_at_bench_size():
    """The same comparison once at BASELINE configs[1]'s tile size (1024 x 1024, batch 1 to keep the fp32 MFMA run short):
    bf16 training kernels against the fp32 parity mode of the same model on the same tile -- FPN maps within 2e-2 relative L2
    (bf16 operands through 53 convs; 256^2 tiles measure 9e-3 against the reference), RPN and total loss within 2 %."""
    from bonai_amd.synth import make_batch
    data = make_batch(1, 1024, 80, device='cuda')
    m = _build()
    with torch.no_grad():
        f16 = [f.float() for f in m.extract_feat(data['img'])]
        l16 = dict(m.train_step(data)['log_vars'].items())
        m.backbone.compute_dtype = torch.float32
        f32 = m.extract_feat(data['img'])
        l32 = dict(m.train_step(data)['log_vars'].items())
    rel = [float((a - b).norm() / b.norm()) for a, b in zip(f16, f32)]
    print('bf16 vs fp32 parity mode at 1024^2, relative L2 per FPN level:', [f'{r:.2e}' for r in rel],
          {k: (round(l16[k], 4), round(l32[k], 4)) for k in l16})
    assert all(f.dtype == torch.float32 for f in f32) and max(rel) < 2e-2, rel
    for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss'):
        assert abs(l16[k] - l32[k]) <= 2e-2 * max(1.0, abs(l32[k])), (k, l16[k], l32[k])
