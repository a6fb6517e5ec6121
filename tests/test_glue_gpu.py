"""GPU parity: HBM-bound glue + box/target kernels vs the CPU restatement (oracle/ops_ref.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cops, ops_ref

pytestmark = pytest.mark.gpu


def _cl(t, dtype=torch.bfloat16):
    return t.to('cuda', dtype).contiguous(memory_format=torch.channels_last)


def _r(t):
    return t.bfloat16().float()


def test_elementwise():
    from bonai_amd import kernels as K
    torch.manual_seed(0)
    g, y = _r(torch.randn(2, 64, 10, 12)), _r(torch.randn(2, 64, 10, 12))
    out = K.relu_bwd(_cl(g), _cl(y)).float().cpu()
    assert torch.equal(out, g * (y > 0))
    cs = K.colsum(_cl(g), 64).cpu()
    assert torch.allclose(cs, g.sum(dim=(0, 2, 3)), atol=1e-3)
    fine, coarse = _r(torch.randn(2, 32, 8, 12)), _r(torch.randn(2, 32, 4, 6))
    up = K.upsample2x_add_(_cl(fine), _cl(coarse)).float().cpu()
    assert torch.allclose(up, _r(fine + F.interpolate(coarse, scale_factor=2, mode='nearest')))
    dn = K.downsum2x_add_(_cl(coarse), _cl(fine)).float().cpu()
    assert torch.allclose(dn, _r(coarse + 4 * F.avg_pool2d(fine, 2)), atol=2e-2)
    x = _r(torch.randn(2, 16, 9, 11))
    assert torch.equal(K.subsample2(_cl(x)).float().cpu(), F.max_pool2d(x, 1, stride=2))
    assert torch.equal(K.maxpool3x3s2(_cl(x)).float().cpu(), F.max_pool2d(x, 3, 2, 1))
    big = _r(torch.randn(2, 16, 8, 8))
    small = _r(torch.randn(2, 16, 4, 4))
    ref = big.clone()
    ref[:, :, ::2, ::2] += small
    assert torch.allclose(K.subsample2_adjoint_add_(_cl(big), _cl(small)).float().cpu(), _r(ref))
    f = torch.randn(4096)
    assert torch.equal(K.cast_bf16(f.cuda()).float().cpu(), _r(f))
    assert torch.equal(K.add_bf16(_cl(g), _cl(y)).float().cpu(), _r(g + y))


def test_stem():
    from bonai_amd import kernels as K
    torch.manual_seed(1)
    img = torch.randn(2, 3, 70, 66)
    w = torch.randn(64, 3, 7, 7) * 0.1
    scale, shift = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    ref = F.relu(F.conv2d(img, w, None, 2, 3) * scale[None, :, None, None] + shift[None, :, None, None])
    out = K.stem7x7_bn_relu(img.cuda(), w.cuda(), scale.cuda(), shift.cuda()).float().cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
    # MFMA stem: bf16 operands (image and folded weights rounded), fp32 accumulate
    imr = img.bfloat16().float()
    wr = (w * scale[:, None, None, None]).bfloat16().float()
    ref2 = F.relu(F.conv2d(imr, wr, None, 2, 3) + shift[None, :, None, None])
    out2 = K.stem7x7_mfma(img.cuda(), w.cuda(), scale.cuda(), shift.cuda()).float().cpu()
    assert out2.shape == ref2.shape
    assert (out2 - ref2).abs().max().item() < 1e-2 * max(1.0, ref2.abs().max().item())


def test_sgd_and_norm():
    from bonai_amd import kernels as K
    torch.manual_seed(2)
    n = 100003
    p, g, m = torch.randn(n), torch.randn(n) * 3, torch.randn(n)
    pd, gd, md = p.cuda(), g.cuda(), m.cuda()
    ss = torch.zeros(1, device='cuda')
    K.sumsq_(gd, ss)
    assert abs(ss.item() - (g.double() ** 2).sum().item()) < 1e-3 * (g.double() ** 2).sum().item()
    K.sgd_momentum_(pd, gd, md, ss, 35.0, 0.005, 0.9, 1e-4)
    norm = g.norm()
    clip = 35.0 / (norm + 1e-6) if norm > 35.0 else 1.0
    d = g * clip + 1e-4 * p
    m2 = 0.9 * m + d
    p2 = p - 0.005 * m2
    assert torch.allclose(md.cpu(), m2, atol=1e-5) and torch.allclose(pd.cpu(), p2, atol=1e-5)


def _boxes(rng, n, size=1024.):
    cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
    w, h = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
    return torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size), dtype=torch.float32)


@pytest.mark.parametrize('thr', [(0.7, 0.3, 0.3), (0.5, 0.5, 0.5)])
def test_iou_assign_bit_exact(thr):
    from bonai_amd import kernels as K
    rng = np.random.RandomState(5)
    B, N, Kmax = 3, 5000, 80
    ngt = [80, 17, 0]
    nbox = [5000, 4321, 100]
    boxes = torch.stack([_boxes(rng, N) for _ in range(B)])
    gts = torch.stack([_boxes(rng, Kmax) for _ in range(B)])
    boxes[0, :80] = gts[0]                 # exact matches (IoU == 1) and duplicated maxima
    boxes[0, 100] = boxes[0, 101]
    gi, mo = K.iou_assign(boxes.cuda(), torch.tensor(nbox).cuda(), gts.cuda(), torch.tensor(ngt).cuda(), *thr)
    for b in range(B):
        ref_gi, ref_mo = ops_ref.max_iou_assign(boxes[b, :nbox[b]], gts[b, :ngt[b]], *thr)
        assert torch.equal(gi[b, :nbox[b]].cpu(), ref_gi), b
        assert torch.equal(mo[b, :nbox[b]].cpu(), ref_mo), b
        assert (gi[b, nbox[b]:] == -1).all()


def test_coders():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(6)
    rois, gt = _boxes(rng, 1000), _boxes(rng, 1000)
    rois[0] = torch.tensor([5., 5., 5., 5.])   # zero-size roi (doctest edge case)
    d = torch.tensor(rng.randn(1000, 4), dtype=torch.float32)
    d[1] = torch.tensor([0., 0., 50., -50.])   # dw/dh clamp
    for means, stds in [((0, 0, 0, 0), (1, 1, 1, 1)), ((0, 0, 0, 0), (.1, .1, .2, .2))]:
        ref = ops_ref.delta2bbox(rois, d, means, stds, (1024, 1024))
        got = K.delta2bbox(rois.cuda(), d.cuda(), means, stds, (1024, 1024)).cpu()
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-3)
        ref = ops_ref.bbox2delta(rois[2:], gt[2:], means, stds)
        got = K.bbox2delta(rois[2:].cuda(), gt[2:].cuda(), means, stds).cpu()
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5)
    # in-tree known answer: delta_xywh_bbox_coder.py:149-162
    r = torch.tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    dd = torch.tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    want = torch.tensor([[0., 0., 1., 1.], [0.1409, 0.1409, 2.8591, 2.8591], [0., 0.3161, 4.1945, 0.6839],
                         [5., 5., 5., 5.]])
    got = K.delta2bbox(r.cuda(), dd.cuda(), (0, 0, 0, 0), (1, 1, 1, 1), (32, 32)).cpu()
    assert torch.allclose(got, want, atol=1e-4)


def test_foa_targets_and_fusion():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(7)
    n = 257
    pb = _boxes(rng, n)
    off = torch.tensor(rng.uniform(-40, 40, (n, 2)), dtype=torch.float32)
    ref = ops_ref.foa_offset_targets([pb], [torch.arange(n)], [off])
    got = K.foa_targets(pb.cuda(), off.cuda()).cpu()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
    pred = torch.tensor(rng.randn(4 * n, 2), dtype=torch.float32)
    pred[0] = 0.0   # polarity of an exactly-zero main prediction is -1 (appendix A.4)
    ref = ops_ref.delta2offset(pb, ops_ref.foa_fuse(pred), max_shape=(1024, 1024))
    got = K.foa_fuse_decode(pred.cuda(), pb.cuda()).cpu()
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-5)


def test_mask_target():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(8)
    Kg, H, W, n = 6, 128, 160, 200
    masks = np.zeros((Kg, H, W), np.uint8)
    for i in range(Kg):
        x1, y1 = rng.randint(0, W - 40), rng.randint(0, H - 40)
        masks[i, y1:y1 + rng.randint(8, 40), x1:x1 + rng.randint(8, 40)] = 1
    masks[5] = (rng.rand(H, W) > 0.5)  # noisy mask
    boxes = _boxes(rng, n, 120.)
    boxes[:, [0, 2]] = boxes[:, [0, 2]].clamp(0, W)
    boxes[:, [1, 3]] = boxes[:, [1, 3]].clamp(0, H)
    idx = torch.tensor(rng.randint(0, Kg, n))
    rois = torch.cat([torch.arange(n, dtype=torch.float32)[:, None], boxes], 1)
    sel = torch.from_numpy(masks)[idx].float()[:, None]
    ref = (cops.roi_align_fwd(sel, rois, 28, 1.0, 0, True).squeeze(1) >= 0.5).float()
    got = K.mask_target(torch.from_numpy(masks).cuda(), boxes.cuda(), idx.cuda(), 28).cpu()
    assert (got != ref).float().mean().item() < 1e-4


def test_mask_target_per_image_list_matches_concatenated():
    """Per-image mask tensors addressed in place (instance address table) == the concatenated form, bit for bit."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(9)
    H, W, n = 96, 128, 300
    parts = [torch.from_numpy((rng.rand(k, H, W) > 0.6).astype(np.uint8)).cuda() for k in (3, 0, 5, 1)]
    boxes = _boxes(rng, n, 90.)
    boxes[:, [0, 2]] = boxes[:, [0, 2]].clamp(0, W)
    boxes[:, [1, 3]] = boxes[:, [1, 3]].clamp(0, H)
    idx = torch.tensor(rng.randint(0, 9, n)).cuda()
    a = K.mask_target(parts, boxes.cuda(), idx, 28)
    b = K.mask_target(torch.cat(parts, 0), boxes.cuda(), idx, 28)
    assert torch.equal(a, b)


def test_fold_pack_multi_equals_per_conv_packing():
    """loft_fold_pack_multi (one launch for all trainable convs of a step) writes bit-identical packings to loft_fold_pack:
    3x3 / 1x1 / 2x2-deconv / 5x5 (per-element form) taps, BN-folded and biased, channel-padded, grouped, with and without the
    transposed (dgrad) packing."""
    from bonai_amd import kernels as K
    torch.manual_seed(5)
    dev = 'cuda'
    reg = K.PrepackRegistry()
    reg.run(0)
    cases = []   # (Cout, Cin, k, bn?, bias?, cout_p, cin_p, dgrad, groups)
    for spec in [(256, 256, 3, True, False, 256, 256, True, 1), (64, 256, 1, True, False, 64, 256, True, 1),
                 (1024, 1000, 1, False, True, 1024, 1000, True, 1), (256, 256, 2, False, True, 256, 256, True, 1),
                 (32, 48, 3, True, False, 64, 64, True, 1), (18, 30, 3, False, False, 18, 30, False, 1),
                 (64, 64, 5, True, False, 64, 64, True, 1), (128, 128, 3, False, True, 128, 128, True, 4),
                 (2, 1024, 1, False, True, 2, 1024, True, 1)]:
        Cout, Cin, k, has_bn, has_b, cop, cip, dgrad, G = spec
        ws = tuple(torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=dev)) for _ in range(G))
        bs = tuple(torch.nn.Parameter(torch.randn(Cout, device=dev)) if has_b else None for _ in range(G))
        bn = tuple(torch.rand(Cout, device=dev) + 0.5 for _ in range(4)) if has_bn else None
        cases.append((ws, bs, bn, cop, cip, dgrad))
        reg.request(ws, bs, bn, 1e-5, cop, cip, dgrad)          # registers (and packs per conv: first step)
    for ws, bs, bn, *_ in cases:                                 # new weights: only the batched launch can produce these
        for w in ws:
            w.data.normal_()
    reg.run(1)
    for ws, bs, bn, cop, cip, dgrad in cases:
        wp, wpt, bias = reg.request(ws, bs, bn, 1e-5, cop, cip, dgrad)
        for g in range(len(ws)):
            rwp, rwpt, rb = K.fold_pack(ws[g], bs[g], bn, 1e-5, want_dgrad=dgrad, cout_pad=cop, cin_pad=cip)
            assert torch.equal(wp[g].view(torch.int16), rwp.view(torch.int16))
            assert torch.equal(bias[g], rb)
            if dgrad:
                assert torch.equal(wpt[g].view(torch.int16), rwpt.view(torch.int16))
            else:
                assert wpt is None


def test_fold_pack_multi_flat_linear_packing():
    """n-major records (a Linear over a flattened [C,H,W] map kept NHWC): the batched launch + transpose give exactly the
    packings of the column-permuted weight, and the batched unpack returns the gradient in the parameter's (c, h, w) order."""
    from bonai_amd import kernels as K
    torch.manual_seed(11)
    dev = 'cuda'
    O, C, H, W = 96, 32, 7, 7
    w = torch.nn.Parameter(torch.randn(O, C * H * W, device=dev))
    b = torch.nn.Parameter(torch.randn(O, device=dev))
    reg = K.PrepackRegistry()
    reg.run(0)
    reg.request((w,), (b,), None, 1e-5, O, C * H * W, True, flat_chw=(C, H, W))
    w.data.normal_()
    reg.run(1)
    wp, wpt, bias = reg.request((w,), (b,), None, 1e-5, O, C * H * W, True, flat_chw=(C, H, W))
    wperm = w.detach().view(O, C, H, W).permute(0, 2, 3, 1).reshape(O, -1, 1, 1).contiguous()
    rwp, rwpt, rb = K.fold_pack(wperm, b, None, 1e-5, want_dgrad=True)
    assert torch.equal(wp[0].view(torch.int16), rwp.view(torch.int16))
    assert torch.equal(wpt[0].view(torch.int16), rwpt.view(torch.int16))
    assert torch.equal(bias[0], rb)
    # unpack: dwp in (h, w, c) order -> accumulate into a (c, h, w)-ordered slot
    dwp = torch.randn(O, H * W * C, device=dev)
    db = torch.randn(O, device=dev)
    slot_w, slot_b = torch.ones(O, C * H * W, device=dev), torch.ones(O, device=dev)
    q = K.UnpackQueue()
    q.add(dwp, db, w, None, 1e-5, (slot_w, None, slot_b), flat_chw=(C, H, W))
    q.flush()
    want = 1.0 + dwp.view(O, H * W, C).permute(0, 2, 1).reshape(O, -1)
    assert torch.equal(slot_w, want)
    assert torch.equal(slot_b, 1.0 + db)


@pytest.mark.parametrize('N,num,frac,npos', [(261888, 256, 0.5, 300), (261888, 256, 0.5, 40), (3080, 1024, 0.25, 600),
                                               (3080, 1024, 0.25, 10), (100, 256, 0.5, 30), (1500, 512, 0.25, 0)])
def test_random_sampler_kernel(N, num, frac, npos):
    """loft_random_sample: 'first' mode equals the torch implementation exactly; 'random' mode draws valid, ascending, unique
    subsets of the right sizes, reproducibly for a seed and differently for another, and roughly uniformly."""
    from bonai_amd.loft.core import RandomSampler
    rng = np.random.RandomState(N + npos)
    B = 4
    gi = np.zeros((B, N), np.int64)
    for b in range(B):
        pos = rng.choice(N, size=min(N, npos + b), replace=False)
        gi[b, pos] = rng.randint(1, 80, size=pos.shape[0])
        ign = rng.choice(N, size=N // 50, replace=False)
        gi[b, ign] = np.where(gi[b, ign] > 0, gi[b, ign], -1)
    gt_inds = torch.from_numpy(gi).cuda()
    smp = RandomSampler(num, frac)
    smp.choice_mode = 'first'
    got, want = smp.sample_batched(gt_inds), smp.sample_batched_host(gt_inds)
    for k in ('pos_valid', 'neg_valid'):
        assert torch.equal(got[k], want[k]), k
    for k, v in (('pos_idx', 'pos_valid'), ('neg_idx', 'neg_valid')):
        assert torch.equal(got[k][got[v]], want[k][want[v]]), k
        assert int(got[k].max()) <= N - 1
    smp.choice_mode = 'random'
    torch.manual_seed(3)
    from bonai_amd import kernels as K
    K._SAMPLE_CALLS[0] = 0
    a = smp.sample_batched(gt_inds)
    K._SAMPLE_CALLS[0] = 0
    a2 = smp.sample_batched(gt_inds)
    c = smp.sample_batched(gt_inds)            # next call: another draw
    P = min(int(num * frac), N)
    for b in range(B):
        np_all = int((gt_inds[b] > 0).sum())
        nn_all = int((gt_inds[b] == 0).sum())
        kp = min(np_all, P)
        kn = min(nn_all, num - kp)
        pv, nv = a['pos_valid'][b], a['neg_valid'][b]
        assert int(pv.sum()) == kp and int(nv.sum()) == kn
        assert bool(pv[:kp].all()) and bool(nv[:kn].all())           # valid slots first
        pi, ni = a['pos_idx'][b][pv], a['neg_idx'][b][nv]
        assert bool((gt_inds[b][pi] > 0).all()) and bool((gt_inds[b][ni] == 0).all())
        assert bool((pi[1:] > pi[:-1]).all()) and bool((ni[1:] > ni[:-1]).all())   # ascending and unique
    for k in a:
        assert torch.equal(a[k], a2[k])
    if N > 3000 and npos >= 40:
        assert not torch.equal(a['neg_idx'], c['neg_idx'])
        # uniformity: mean of the drawn negative indices ~ N/2 (std of the mean of 256 uniform draws = N / sqrt(12 * 256))
        m = a['neg_idx'][a['neg_valid']].float().mean().item()
        assert abs(m - N / 2) < 6 * N / (12 * a['neg_valid'].sum().item()) ** 0.5


def test_fused_losses_match_elementwise_formulation():
    """loft_fused_loss (value + gradient in one launch) against the torch formulation of the same modules: L1, SmoothL1,
    sigmoid / softmax / mask cross-entropy with weights, device and python avg_factor, plain mean."""
    import os
    from bonai_amd.loft.losses import CrossEntropyLoss, L1Loss, SmoothL1Loss
    torch.manual_seed(0)
    dev = 'cuda'
    cases = []
    p = torch.randn(8, 128, 4, device=dev)
    cases.append((L1Loss(loss_weight=1.0), (p, torch.randn_like(p), (torch.rand(8, 128, 1, device=dev) > 0.5).float().expand_as(p)),
                  dict(avg_factor=torch.tensor(1900.0, device=dev))))
    p = torch.randn(4000, 4, device=dev) * 2
    cases.append((SmoothL1Loss(beta=1.0, loss_weight=1.0), (p, torch.randn_like(p), (torch.rand(4000, 4, device=dev) > 0.7).float()),
                  dict(avg_factor=4000.0)))
    p = torch.randn(3000, 2, device=dev)
    cases.append((SmoothL1Loss(beta=1.0, loss_weight=16.0), (p, torch.randn_like(p)), dict()))
    p = torch.randn(3072, 1, device=dev) * 3
    cases.append((CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0), (p, torch.randint(0, 2, (3072,), device=dev),
                                                                         (torch.rand(3072, device=dev) > 0.2).float()),
                  dict(avg_factor=torch.tensor(2450.0, device=dev))))
    p = torch.randn(8192, 2, device=dev) * 2
    cases.append((CrossEntropyLoss(loss_weight=1.0), (p, torch.randint(0, 2, (8192,), device=dev), torch.ones(8192, device=dev)),
                  dict(avg_factor=torch.tensor(8192.0, device=dev))))
    p = torch.randn(300, 1, 28, 28, device=dev) * 2
    cases.append((CrossEntropyLoss(use_mask=True, loss_weight=1.0), (p, (torch.rand(300, 28, 28, device=dev) > 0.5).float(),
                                                                      torch.zeros(300, dtype=torch.long, device=dev)), dict()))
    for mod, args, kw in cases:
        res = []
        from bonai_amd.loft import losses as LS
        for elementwise in (False, True):
            LS.ELEMENTWISE_ONLY = elementwise
            try:
                pred = args[0].clone().requires_grad_(True)
                loss = mod(pred, *args[1:], **kw)
                (loss.sum() * 1.7).backward()
                res.append((loss.detach().reshape(-1), pred.grad.clone()))
            finally:
                LS.ELEMENTWISE_ONLY = False
        (l0, g0), (l1, g1) = res
        assert l0.shape == l1.shape
        assert (l0 - l1).abs().max().item() <= 2e-5 * max(1.0, l1.abs().max().item()), type(mod).__name__
        assert (g0 - g1).abs().max().item() <= 2e-5 * max(1e-6, g1.abs().max().item()), type(mod).__name__


def test_fused_losses_read_strided_views_bool_weights_and_count_accuracy():
    """loft_fused_loss_v2: the operands where the heads leave them -- class / delta columns of the fused [n, 8] fc_cls + fc_reg
    output, rows b, p < P and columns 1..4 of the RPN's [B, S, 5] gather with the positives' bool validity as the row weight, one
    channel of a 4-padded NHWC mask-logit map -- against the elementwise formulation on contiguous copies; the top-1 accuracy
    counted by the softmax launch (accuracy.py:4-48)."""
    from bonai_amd import kernels as K
    from bonai_amd.loft import losses as LS
    from bonai_amd.loft.losses import CrossEntropyLoss, L1Loss, SmoothL1Loss, accuracy
    torch.manual_seed(1)
    dev = 'cuda'

    def both(mod, make_pred, rest, kw):
        out = []
        for elementwise in (False, True):
            LS.ELEMENTWISE_ONLY = elementwise
            try:
                base, view = make_pred()
                a = [t.float() if (elementwise and t.dtype == torch.bool) else t for t in rest]
                loss = mod(view.contiguous() if elementwise else view, *a, **kw)
                (loss.sum() * 1.3).backward()
                out.append((loss.detach().reshape(-1), base.grad.clone()))
            finally:
                LS.ELEMENTWISE_ONLY = False
        (l0, g0), (l1, g1) = out
        assert (l0 - l1).abs().max().item() <= 2e-5 * max(1.0, l1.abs().max().item())
        assert (g0 - g1).abs().max().item() <= 2e-5 * max(1e-6, g1.abs().max().item())
        return out

    o8 = torch.randn(4096, 8, device=dev) * 2

    def head_cols(lo, hi):
        def f():
            b = o8.clone().requires_grad_(True)
            return b, b[:, lo:hi]
        return f
    lab = torch.randint(0, 2, (4096,), device=dev)
    ce = CrossEntropyLoss(loss_weight=1.0)
    both(ce, head_cols(0, 2), (lab, torch.ones(4096, device=dev)), dict(avg_factor=4096.0))
    # accuracy: counted by the same launch, handed out once for this very (pred, label) pair
    b = o8.clone()
    v = b[:, :2]
    ce(v, lab, torch.ones(4096, device=dev), avg_factor=4096.0)
    acc = accuracy(v, lab, loss_module=ce)
    want = (v.argmax(1) == lab).float().sum() * (100.0 / 4096)
    assert abs(float(acc) - float(want)) < 1e-3 and ce.last_accuracy is None
    assert abs(float(accuracy(v, lab, loss_module=ce)) - float(want)) < 1e-3            # second call: the plain formulation
    tie = torch.zeros(64, 2, device=dev)                                                # ties -> first maximum (torch.argmax)
    tl = torch.randint(0, 2, (64,), device=dev)
    ce(tie, tl, None, avg_factor=64.0)
    assert abs(float(accuracy(tie, tl, loss_module=ce)) - float((tl == 0).float().mean() * 100)) < 1e-3
    both(L1Loss(loss_weight=1.0), head_cols(2, 6), (torch.randn(4096, 4, device=dev), (torch.rand(4096, 4, device=dev) > 0.5).float()),
         dict(avg_factor=4096.0))
    vals = torch.randn(8, 768, 5, device=dev)

    def rpn_pred():
        b = vals.clone().requires_grad_(True)
        return b, b[:, :256, 1:5]

    def rpn_logit():
        b = vals.clone().requires_grad_(True)
        return b, b[..., 0].reshape(-1, 1)
    pval = torch.rand(8, 256, device=dev) > 0.4
    avg = torch.tensor(3000.0, device=dev)
    both(L1Loss(loss_weight=1.0), rpn_pred, (torch.randn(8, 256, 4, device=dev), pval[..., None].expand(8, 256, 4)), dict(avg_factor=avg))
    both(CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0), rpn_logit,
         (torch.randint(0, 2, (6144,), device=dev), (torch.rand(6144, device=dev) > 0.1).float()), dict(avg_factor=avg))
    m4 = torch.randn(120, 4, 28, 28, device=dev).contiguous(memory_format=torch.channels_last)

    def mask_logit():
        b = m4.clone(memory_format=torch.preserve_format).requires_grad_(True)
        return b, b[:, :1]
    both(CrossEntropyLoss(use_mask=True, loss_weight=1.0), mask_logit,
         ((torch.rand(120, 28, 28, device=dev) > 0.5).float(), torch.zeros(120, dtype=torch.long, device=dev)), dict())
    both(SmoothL1Loss(beta=1.0, loss_weight=16.0), head_cols(6, 8), (torch.randn(4096, 2, device=dev),), dict())


def test_roi_sample_targets_matches_tensor_formulation():
    """loft_roi_sample_targets (sampled RoIs per image [pos..., neg...], labels, bbox2delta targets, positives' lists) against the
    gather / nonzero formulation it replaces, incl. an image without positives and one without negatives."""
    from bonai_amd import kernels as K
    from bonai_amd.loft.core import RandomSampler
    rng = np.random.RandomState(2)
    B, N, Kmax, num = 4, 700, 9, 128
    x1, y1 = rng.uniform(0, 400, (B, N)), rng.uniform(0, 400, (B, N))
    cand = torch.tensor(np.stack([x1, y1, x1 + rng.uniform(8, 100, (B, N)), y1 + rng.uniform(8, 100, (B, N))], -1), dtype=torch.float32).cuda()
    gx, gy = rng.uniform(0, 400, (B, Kmax)), rng.uniform(0, 400, (B, Kmax))
    gts = torch.tensor(np.stack([gx, gy, gx + rng.uniform(8, 100, (B, Kmax)), gy + rng.uniform(8, 100, (B, Kmax))], -1), dtype=torch.float32).cuda()
    gt_inds = torch.tensor(rng.choice([-1, 0, 0, 0, 1, 2, 5, 9], size=(B, N)), dtype=torch.int64)
    gt_inds[1] = gt_inds[1].clamp(max=0)          # image without positives
    gt_inds[2] = gt_inds[2].clamp(min=1)          # image without negatives
    gt_inds = gt_inds.cuda()
    lab = torch.tensor(rng.randint(0, 3, (B, Kmax)), dtype=torch.int64).cuda()
    smp = RandomSampler(num, 0.25)
    smp.choice_mode = 'first'
    r = smp.sample_batched(gt_inds)
    pidx, pval, nidx, nval = r['pos_idx'], r['pos_valid'], r['neg_idx'], r['neg_valid']
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    got = K.roi_sample_targets(cand, gt_inds, gts, lab, pidx, pval, nidx, nval, 3, means, stds)
    # the tensor formulation
    idx, val = torch.cat([pidx, nidx], 1), torch.cat([pval, nval], 1)
    is_pos = torch.cat([pval, torch.zeros_like(nval)], 1)
    bidx = torch.arange(B, device='cuda')[:, None].expand_as(idx)
    sel = val.reshape(-1).nonzero().flatten()
    b_s, i_s, pos_s = bidx.reshape(-1)[sel], idx.reshape(-1)[sel], is_pos.reshape(-1)[sel]
    rois = torch.cat([b_s[:, None].float(), cand[b_s, i_s]], 1)
    assigned = (gt_inds[b_s, i_s] - 1).clamp(min=0)
    pos_sel = pos_s.nonzero().flatten()
    labels = torch.full((rois.shape[0],), 3, dtype=torch.long, device='cuda')
    labels[pos_sel] = lab[b_s[pos_sel], assigned[pos_sel]]
    tgt = torch.zeros(rois.shape[0], 4, device='cuda')
    tgt[pos_sel] = K.bbox2delta(rois[pos_sel][:, 1:].contiguous(), gts[b_s[pos_sel], assigned[pos_sel]], means, stds)
    assert torch.equal(got['rois'], rois) and torch.equal(got['labels'], labels)
    assert torch.equal(got['pos_sel'], pos_sel) and torch.equal(got['pos_b'], b_s[pos_sel]) and torch.equal(got['pos_gt_i'], assigned[pos_sel])
    assert torch.equal(got['pos_rois'], rois[pos_sel])
    assert torch.equal(got['bbox_targets'], tgt)
    w = torch.zeros(rois.shape[0], 4, device='cuda')
    w[pos_sel] = 1.0
    assert torch.equal(got['bbox_weights'], w) and bool((got['label_weights'] == 1).all())
    # the two-phase form (launch without the counts, read them later): the list a full sampler would produce is available at
    # once -- here two images fall short, so it carries the real rows first and zero boxes behind them
    h = K.roi_sample_targets_begin(cand, gt_inds, gts, lab, pidx, pval, nidx, nval, 3, means, stds, num_expected=num)
    early = h.rois_max.clone()
    fin = h.finish()
    M = rois.shape[0]
    assert M < B * num and early.shape[0] == B * num
    assert torch.equal(early[:M], rois) and bool((early[M:] == 0).all())
    assert all(torch.equal(fin[k], got[k]) for k in got)


def test_premasked_gradient_with_two_consumers_any_order():
    """ADVICE r1 (nn.py pre-masked gradients): y = relu(conv(x)) feeds a consumer that folds y's ReLU mask into its own dgrad
    epilogue (input_relu=True) AND a consumer that does not.  Autograd sums the two contributions in place into whichever arrives
    first; the producer may skip its relu_bwd only when the tagged gradient is the whole gradient.  Both creation orders, against
    the same graph without any epilogue folding and against torch-CPU autograd."""
    from bonai_amd import nn as F2
    torch.manual_seed(0)
    B, C, H, W = 2, 128, 12, 12
    x0 = torch.randn(B, C, H, W).bfloat16()
    ws = [torch.nn.Parameter((torch.randn(C, C, 3, 3) * (1.0 / (C * 9)) ** 0.5).cuda()) for _ in range(3)]

    def run(order, fold):
        for w in ws:
            w.grad = None
        x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = F2.conv2d(x, ws[0], None, pad=1, relu=True)
        if order == 'plain_first':
            b = F2.conv2d(y, ws[2], None, pad=1, relu=False)                       # consumer without mask folding
            a = F2.conv2d(y, ws[1], None, pad=1, relu=True, input_relu=fold)       # consumer that folds y's mask
        else:
            a = F2.conv2d(y, ws[1], None, pad=1, relu=True, input_relu=fold)
            b = F2.conv2d(y, ws[2], None, pad=1, relu=False)
        (a.float().sum() * 0.5 + (b.float() * b.float()).sum() * 0.1).backward()
        return x.grad.float().cpu(), [w.grad.float().cpu().clone() for w in ws]

    ref_x, ref_w = run('plain_first', fold=False)
    for order in ('plain_first', 'folding_first'):
        gx, gw = run(order, fold=True)
        assert (gx - ref_x).abs().max().item() <= 2e-2 * ref_x.abs().max().item(), order
        for g, r in zip(gw, ref_w):
            assert (g - r).abs().max().item() <= 2e-2 * r.abs().max().item(), order
    # and the graph itself against torch-CPU fp32 autograd on the same bf16-rounded operands
    xc = x0.float().requires_grad_(True)
    wc = [w.detach().float().cpu().bfloat16().float().requires_grad_(True) for w in ws]
    yc = torch.relu(torch.nn.functional.conv2d(xc, wc[0], padding=1))
    ac = torch.relu(torch.nn.functional.conv2d(yc, wc[1], padding=1))
    bc = torch.nn.functional.conv2d(yc, wc[2], padding=1)
    (ac.sum() * 0.5 + (bc * bc).sum() * 0.1).backward()
    assert (ref_x - xc.grad).abs().max().item() <= 5e-2 * xc.grad.abs().max().item()
    # a single folding consumer still lets the producer skip its pass (the tag is honoured: exactly one registered use)
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F2.conv2d(x, ws[0], None, pad=1, relu=True)
    a = F2.conv2d(y, ws[1], None, pad=1, relu=True, input_relu=True)
    assert F2._USES.get(y.data_ptr()) == 1


def test_fused_target_and_gather_launches_match_tensor_formulations_in_the_model():
    """The RoI head's sampled lists / targets (loft_roi_sample_targets) and the RPN's sampled-anchor gather
    (loft_rpn_sample_gather) against the tensor formulations they replaced, selected in-process on the whole training step:
    identical sampling ('first-k'), so every loss must agree to fp32 rounding of the 16-bit activations."""
    import os
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector, roi as ROI, rpn as RPN
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(root, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    torch.manual_seed(0)
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    data = make_batch(2, 256, 6, device='cuda')
    res = []
    for flags in ((False, False), (True, True)):
        ROI.TENSOR_TARGETS, RPN.TENSOR_GATHER = flags
        try:
            res.append({k: float(v) for k, v in m.train_step(data)['log_vars'].items()})
        finally:
            ROI.TENSOR_TARGETS = RPN.TENSOR_GATHER = False
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) <= 2e-3 * max(1.0, abs(res[1][k])), (k, res[0][k], res[1][k])


def test_fused_roi_backward_matches_per_extractor_passes():
    """The fused RoIAlign backward of the three extractors (nn._hub_flush -> loft_roi_align_bwd_multi) against one accumulate
    pass per extractor: same data, same sampling -> the arena gradients agree to the 16-bit rounding of the shared maps."""
    import os
    from bonai_amd import nn as F2
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(root, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    data = make_batch(2, 256, 8, device='cuda')

    def run(fused):
        torch.manual_seed(0)
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        tr = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0, max_norm=0.0)
        F2.ROI_BWD_FUSED = fused
        try:
            tr.train_step(data)
        finally:
            F2.ROI_BWD_FUSED = True
        return tr.arena.grad.clone()
    ref, got = run(False), run(True)
    assert torch.isfinite(got).all()
    assert (got - ref).norm().item() <= 1e-2 * ref.norm().item(), ((got - ref).norm().item(), ref.norm().item())


def test_product_anchor_generator_device_tables_vs_reference_fixture(golden_dir):
    """a6: the anchor tables the RPN head decodes from ON THE DEVICE (AnchorGenerator.grid_anchors(device='cuda') and the
    base-anchor tables handed to loft_rpn_decode) are bit-identical to mmdet's own generator (anchor_generator.py:142-265)."""
    import os
    from bonai_amd.loft.builder import build_anchor_generator
    g = np.load(os.path.join(golden_dir, 'core_ops.npz'))
    ag = build_anchor_generator(dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]))
    sizes = [tuple(int(v) for v in s) for s in g['anchor_sizes']]
    got = ag.grid_anchors(sizes, device='cuda')
    for i, a in enumerate(got):
        assert a.is_cuda and torch.equal(a.cpu(), torch.from_numpy(g[f'anchors_{i}'])), i
    assert ag.grid_anchors(sizes, device='cuda') is got                 # cached per (sizes, device)
    # the decode kernel rebuilds anchor k of cell (y, x) as base[k] + (x, y, x, y) * stride: same fp32 values as the table
    for (h, w), s, ba, tab in zip(sizes, ag.strides, ag.base_anchors, got):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
        sh = torch.stack([xs, ys, xs, ys], -1).view(-1, 1, 4) * s
        assert torch.equal((ba[None] + sh).view(-1, 4), tab.cpu())


def test_small_fused_launches_match_their_tensor_formulations():
    """Three single launches that replaced chains of stock elementwise launches, each against the chain it
    replaced: the sparse RPN backward's operand build (cat / arange / zeros / scatter_ / casts / permuted copies), the RPN losses'
    normaliser (anchor_head.py:363-364) and the out-of-place FPN gradient sum (bit-identical to the in-place form)."""
    from bonai_amd import kernels as K
    torch.manual_seed(3)
    dev = 'cuda'
    dt = K.L.act16()
    A, P, C, nsel = 3, 128, 256, 700
    g = torch.randn(nsel, 5, device=dev)
    slot = torch.randint(0, A, (nsel,), device=dev)
    w_cls, w_reg, w_conv = torch.randn(A, C, 1, 1, device=dev), torch.randn(4 * A, C, 1, 1, device=dev), torch.randn(C, C, 3, 3, device=dev)
    g_rows, w_headT, wd = K.rpn_sparse_prep(g, slot, A, P, w_cls, w_reg, w_conv)
    idx = torch.cat([slot[:, None], A + 4 * slot[:, None] + torch.arange(4, device=dev)[None]], 1)
    want_rows = torch.zeros(nsel, P, device=dev).scatter_(1, idx, g).to(dt)
    w_head = torch.zeros(P, C, device=dev)
    w_head[:A] = w_cls.view(A, C)
    w_head[A:5 * A] = w_reg.view(4 * A, C)
    assert g_rows.dtype == dt and torch.equal(g_rows, want_rows)
    assert torch.equal(w_headT, w_head.t().contiguous().to(dt))
    assert torch.equal(wd, w_conv.permute(2, 3, 1, 0).reshape(9 * C, C).to(dt).contiguous())
    # normaliser: an image without positives and one without negatives count as one each
    pval = torch.rand(6, 128, device=dev) > 0.5
    nval = torch.rand(6, 256, device=dev) > 0.3
    pval[2] = False
    nval[4] = False
    avg = K.sampled_avg_factor(pval, nval)
    want = pval.sum(1).clamp(min=1).sum() + nval.sum(1).clamp(min=1).sum()
    assert avg.dtype == torch.float32 and float(avg) == float(want)
    # FPN gradient sum
    coarse = torch.randn(2, 256, 24, 40, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    fine = torch.randn(2, 256, 48, 80, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    keep = coarse.clone(memory_format=torch.preserve_format)
    out = K.downsum2x_sum(coarse, fine)
    assert torch.equal(coarse, keep) and out.data_ptr() != coarse.data_ptr()
    assert torch.equal(out, K.downsum2x_add_(coarse.clone(memory_format=torch.preserve_format), fine))
