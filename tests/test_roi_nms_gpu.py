"""GPU parity: HIP RoIAlign / NMS / sort vs the plain-C oracle (oracle/loft_oracle.c)."""
import numpy as np
import pytest
import torch

from oracle import cops, ops_ref

pytestmark = pytest.mark.gpu


def _rand_rois(rng, n, B, size, wmin=4, wmax=400):
    w = np.exp(rng.uniform(np.log(wmin), np.log(wmax), n))
    h = np.exp(rng.uniform(np.log(wmin), np.log(wmax), n))
    x1 = rng.uniform(-20, size - 8, n)
    y1 = rng.uniform(-20, size - 8, n)
    b = rng.randint(0, B, n)
    return torch.tensor(np.stack([b, x1, y1, x1 + w, y1 + h], 1), dtype=torch.float32)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('P,n_rot', [(7, 1), (14, 1), (7, 4)])
def test_roi_align_fwd_bwd(dtype, P, n_rot):
    from bonai_amd import kernels as K
    rng = np.random.RandomState(0)
    B, C, size = 2, 64, 256
    strides = [4, 8, 16, 32]
    torch.manual_seed(0)
    feats = [torch.randn(B, C, size // s, size // s) for s in strides]
    if dtype == torch.bfloat16:
        feats = [f.bfloat16().float() for f in feats]  # oracle sees the same rounded inputs
    rois = _rand_rois(rng, 300, B, size)
    # edge cases: zero-size box, box fully outside, box covering the image
    rois[0, 1:] = torch.tensor([10., 10., 10., 10.])
    rois[1, 1:] = torch.tensor([-500., -500., -400., -400.])
    rois[2, 1:] = torch.tensor([0., 0., float(size), float(size)])
    ref = ops_ref.roi_extract(feats, rois, P, strides)
    dfeats = [f.to('cuda', dtype).contiguous(memory_format=torch.channels_last) for f in feats]
    out = K.roi_align_fwd(dfeats, rois.cuda(), P, strides, n_rot=n_rot)
    assert out.shape == (n_rot * rois.shape[0], C, P, P)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    for r in range(n_rot):
        got = out[r * rois.shape[0]:(r + 1) * rois.shape[0]].float().cpu()
        want = torch.rot90(ref, r, (2, 3))
        err = (got - want).abs().max().item()
        assert err <= tol * max(1.0, want.abs().max().item()), (r, err)
    # level map
    lv = K.map_roi_levels(rois.cuda()).cpu().long()
    assert torch.equal(lv, ops_ref.map_roi_levels(rois))
    # backward
    g = torch.randn(n_rot * rois.shape[0], C, P, P)
    if dtype == torch.bfloat16:
        g = g.bfloat16().float()
    gsum = sum(torch.rot90(g[r * rois.shape[0]:(r + 1) * rois.shape[0]], -r, (2, 3)) for r in range(n_rot))
    lv = ops_ref.map_roi_levels(rois)
    want_grads = []
    for i, s in enumerate(strides):
        m = lv == i
        want_grads.append(cops.roi_align_bwd(gsum[m], rois[m], feats[i].shape, 1.0 / s) if m.any()
                          else torch.zeros_like(feats[i]))
    got_grads = K.roi_align_bwd(g.to('cuda', dtype).contiguous(memory_format=torch.channels_last), rois.cuda(),
                                [tuple(f.shape) for f in feats], P, strides, n_rot=n_rot)
    for gg, ww in zip(got_grads, want_grads):
        err = (gg.cpu() - ww).abs().max().item()
        assert err <= 2e-4 * max(1.0, ww.abs().max().item()), err


def test_roi_align_empty():
    from bonai_amd import kernels as K
    f = [torch.randn(1, 8, 16, 16, device='cuda').contiguous(memory_format=torch.channels_last)]
    out = K.roi_align_fwd(f, torch.zeros(0, 5, device='cuda'), 7, [4])
    assert out.shape == (0, 8, 7, 7)


def _rand_boxes(rng, n, size=1024.):
    cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
    w, h = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size)
    return torch.tensor(b, dtype=torch.float32)


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 1000, 12768])
def test_nms_bit_exact(n):
    from bonai_amd import kernels as K
    rng = np.random.RandomState(n)
    boxes = _rand_boxes(rng, n)
    # quantised scores force many exact ties -> exercises the (score desc, index asc) order
    scores = torch.tensor(np.round(rng.uniform(0, 1, n), 2), dtype=torch.float32)
    if n > 10:
        boxes[5] = boxes[3]          # identical boxes
        scores[5] = scores[3]
    dets_ref, keep_ref = cops.nms(boxes, scores, 0.7)
    dets, keep = K.nms(boxes.cuda(), scores.cuda(), 0.7)
    assert torch.equal(keep.cpu(), keep_ref)
    assert torch.equal(dets.cpu(), dets_ref)


def test_nms_segmented_matches_batched():
    """Segment-wise NMS + global re-sort == mmcv batched_nms on the shifted boxes."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(7)
    sizes = [3000, 3000, 1500, 700, 0, 64]
    boxes = torch.cat([_rand_boxes(rng, s) for s in sizes])
    scores = torch.tensor(np.round(rng.uniform(0, 1, boxes.shape[0]), 3), dtype=torch.float32)
    ids = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
    dets_ref, keep_ref = cops.batched_nms(boxes, scores, ids, dict(type='nms', iou_threshold=0.7))
    off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64).cuda()
    ks, order = K.segmented_sort_desc(scores.cuda(), off)
    order = order.long()
    shift = (torch.arange(len(sizes), dtype=torch.float32) * (boxes.max() + 1)).cuda()
    keep_mask = K.nms_segmented(boxes.cuda()[order], off, 0.7, seg_shift=shift)
    kept = order[keep_mask.bool()]
    # global order: score desc, original index asc
    kept_cpu = kept.cpu()
    kk = sorted(kept_cpu.tolist(), key=lambda i: (-scores[i].item(), i))
    assert kk == keep_ref.tolist()


def _tie_boxes(rng, n):
    """Boxes on a small integer lattice: many pairs whose IoU equals 0.5 EXACTLY (e.g. [0,0,1,2] vs [0,0,1,1]: inter 1, union 2),
    which is where the two mmcv-1.0.5 predicates differ (device: inter > thr*union keeps; host: inter/union >= thr suppresses)."""
    x1, y1 = rng.randint(0, 6, n), rng.randint(0, 6, n)
    w, h = rng.randint(1, 4, n), rng.randint(1, 4, n)
    return torch.tensor(np.stack([x1, y1, x1 + w, y1 + h], 1), dtype=torch.float32)


@pytest.mark.parametrize('predicate', ['device', 'cpu'])
def test_nms_predicate_at_threshold_ties(predicate):
    """DESIGN.md 'NMS predicate': both published mmcv-1.0.5 predicates, bit-exact against the oracle, on inputs where they
    differ; plus the two-box known answer."""
    from bonai_amd import kernels as K
    two = torch.tensor([[0., 0., 1., 2.], [0., 0., 1., 1.]])
    sc2 = torch.tensor([0.9, 0.8])
    _, keep = K.nms(two.cuda(), sc2.cuda(), 0.5, predicate=predicate)
    assert keep.cpu().tolist() == ([0, 1] if predicate == 'device' else [0])          # IoU == thr exactly
    rng = np.random.RandomState(11)
    differ = 0
    for n in (64, 500, 3000):
        boxes = _tie_boxes(rng, n)
        scores = torch.tensor(rng.permutation(n).astype(np.float32) / n)
        _, kref = cops.nms(boxes, scores, 0.5, predicate=predicate)
        _, kother = cops.nms(boxes, scores, 0.5, predicate='cpu' if predicate == 'device' else 'device')
        differ += int(kref.tolist() != kother.tolist())
        _, keep = K.nms(boxes.cuda(), scores.cuda(), 0.5, predicate=predicate)
        assert keep.cpu().tolist() == kref.tolist()
    assert differ >= 2          # the inputs do exercise the difference
    # random float boxes: the two forms agree except (rarely) within one rounding of the threshold; each is bit-exact to its oracle
    boxes = _rand_boxes(rng, 4000)
    scores = torch.tensor(rng.uniform(0, 1, 4000), dtype=torch.float32)
    _, kref = cops.nms(boxes, scores, 0.7, predicate=predicate)
    _, keep = K.nms(boxes.cuda(), scores.cuda(), 0.7, predicate=predicate)
    assert keep.cpu().tolist() == kref.tolist()


def test_sort_stability():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(3)
    n = 200000
    keys = torch.tensor(np.round(rng.randn(n), 1), dtype=torch.float32)
    off = torch.tensor([0, 70000, 70000, n], dtype=torch.int64)
    ks, vs = K.segmented_sort_desc(keys.cuda(), off.cuda())
    for a, b in zip(off[:-1].tolist(), off[1:].tolist()):
        want = cops.argsort_desc(keys[a:b]) + a
        assert torch.equal(vs[a:b].cpu().long(), want)


@pytest.mark.parametrize('k', [1, 768, 1000, 3000, 4096])
def test_segmented_topk_is_the_head_of_the_stable_descending_sort(k):
    """loft_segmented_topk_desc (in-house radix select + LDS bitonic sort; the RPN's `sort(descending)[:nms_pre]` and the post-NMS
    `[:nms_post]`): for every segment the first min(k, length) keys AND indices equal the stable descending sort's -- heavy ties
    (keys quantised to one decimal: thousands of equal keys straddling position k), +0.0 / -0.0, segments of length 0, 1, < k,
    = k, the 3 x 256^2 anchors of a P2 level, an all-equal segment, suppressed (-1) tails as after NMS."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(k)
    lens = [0, 1, 5, 768, 3000, 3072, 12768, 196608, 4096, 9000, 50000]
    off = np.concatenate([[0], np.cumsum(lens)])
    n = int(off[-1])
    keys = np.round(rng.randn(n), 1).astype(np.float32)
    keys[off[8]:off[9]] = 0.25                                           # all equal
    seg9 = keys[off[9]:off[10]]
    seg9[rng.rand(seg9.size) < 0.9] = -1.0                               # mostly suppressed, like the masked post-NMS scores
    keys[off[10]:off[10] + 100] = -0.0
    keys[off[10] + 100:off[10] + 200] = 0.0
    fine = rng.rand(lens[6]).astype(np.float32)                          # a segment without ties
    keys[off[6]:off[7]] = fine
    kt = torch.from_numpy(keys)
    K.TOPK_MAX_SEGMENT, keep = 1 << 30, K.TOPK_MAX_SEGMENT          # (every segment through the in-house kernel, also the 196 608-key one)
    ks, vs = K.segmented_topk_desc(kt.cuda(), torch.from_numpy(off).cuda(), k, max_segment=max(lens))
    ks, vs = ks.cpu(), vs.cpu().long()
    for a, b in zip(off[:-1].tolist(), off[1:].tolist()):
        kk = min(k, b - a)
        if kk == 0:
            continue
        canon = torch.where(kt[a:b] == 0, torch.zeros(()), kt[a:b])        # the sort treats -0.0 as +0.0 (and returns +0.0)
        order = torch.sort(canon, descending=True, stable=True)[1][:kk]
        assert torch.equal(vs[a:a + kk], order + a), (a, b, kk)
        assert torch.equal(ks[a:a + kk], canon[order]), (a, b, kk)
    # with explicit values, and the k > 4096 fallback = the full sort
    vals = torch.arange(n, dtype=torch.int32).flip(0).contiguous()
    ks2, vs2 = K.segmented_topk_desc(kt.cuda(), torch.from_numpy(off).cuda(), k, values=vals.cuda(), max_segment=max(lens))
    # keep flags as a key mask == where(mask, key, -1) first
    mk = torch.from_numpy(rng.rand(n) < 0.6)
    km, vm = K.segmented_topk_desc(kt.cuda(), torch.from_numpy(off).cuda(), k, max_segment=max(lens), key_mask=mk.cuda())
    kw, vw = K.segmented_topk_desc(torch.where(mk, kt, torch.tensor(-1.0)).cuda(), torch.from_numpy(off).cuda(), k, max_segment=max(lens))
    for a, b in zip(off[:-1].tolist(), off[1:].tolist()):
        kk = min(k, b - a)
        assert torch.equal(km[a:a + kk], kw[a:a + kk]) and torch.equal(vm[a:a + kk], vw[a:a + kk]), (a, b, kk)
    K.TOPK_MAX_SEGMENT = keep
    a, b = int(off[6]), int(off[7])
    kk = min(k, b - a)
    assert torch.equal(vs2[a:a + kk].cpu().long(), vals[vs[a:a + kk]].long())
    fk, fv = K.segmented_topk_desc(kt.cuda(), torch.from_numpy(off).cuda(), 5000, max_segment=max(lens))
    sk, sv = K.segmented_sort_desc(kt.cuda(), torch.from_numpy(off).cuda())
    assert torch.equal(fk, sk) and torch.equal(fv, sv)
    # the TWO-STAGE form the RPN uses for segments past TOPK_MAX_SEGMENT (sub-range top-k -> sorted runs -> rank merge): same heads
    if k > 1:
        assert K._topk_two_stage_tables(lens, k, 'cuda') is not None
        k2, v2 = K.segmented_topk_desc(kt.cuda(), torch.from_numpy(off).cuda(), k, max_segment=max(lens), seg_lengths=lens)
        k2, v2 = k2.cpu(), v2.cpu().long()
        for a, b in zip(off[:-1].tolist(), off[1:].tolist()):
            kk = min(k, b - a)
            assert torch.equal(v2[a:a + kk], vs[a:a + kk]) and torch.equal(k2[a:a + kk], ks[a:a + kk]), (a, b, kk)


@pytest.mark.parametrize('P,n_rot', [(7, 1), (14, 1), (7, 4)])
def test_roi_align_bwd_mfma_matches_scalar_form(P, n_rot):
    """The matrix-core backward (bf16 maps, C = 256: the training path) against the scalar fp32 form that is pinned to the
    oracle above: same RoIs incl. zero-size / outside / image-sized boxes, tiny (sub-pixel-bin) and elongated boxes, and the
    accumulate mode used by the feature-gradient hub."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(5)
    B, C, size = 2, 256, 256
    strides = [4, 8, 16, 32]
    rois = _rand_rois(rng, 400, B, size)
    rois[0, 1:] = torch.tensor([10., 10., 10., 10.])
    rois[1, 1:] = torch.tensor([-500., -500., -400., -400.])
    rois[2, 1:] = torch.tensor([0., 0., float(size), float(size)])
    rois[3, 1:] = torch.tensor([40., 40., 44., 43.])            # sub-pixel bins at stride 4
    rois[4, 1:] = torch.tensor([-20., 100., 250., 112.])        # long and thin, partly outside
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous()
    torch.manual_seed(1)
    g = torch.randn(n_rot * rois.shape[0], C, P, P).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    shapes = [(B, C, size // s, size // s) for s in strides]
    want = K.roi_align_bwd(g, rois.cuda(), shapes, P, strides, n_rot=n_rot, rois_sorted=True, out_dtype=torch.float32)
    got = K.roi_align_bwd(g, rois.cuda(), shapes, P, strides, n_rot=n_rot, rois_sorted=True, out_dtype=torch.bfloat16)
    for gg, ww in zip(got, want):
        assert gg.dtype == torch.bfloat16
        scale = max(1.0, ww.abs().max().item())
        assert (gg.float() - ww).abs().max().item() <= 6e-3 * scale          # bf16 output rounding (2^-8 of the value)
        assert (gg.float() - ww).abs().mean().item() <= 6e-4 * scale
    twice = K.roi_align_bwd(g, rois.cuda(), shapes, P, strides, n_rot=n_rot, rois_sorted=True, grad_feats=[x.clone() for x in got])
    for tt, ww in zip(twice, want):
        scale = max(1.0, ww.abs().max().item())
        assert (tt.float() - 2 * ww).abs().max().item() <= 2e-2 * scale


@pytest.mark.parametrize('P,n_rot', [(7, 1), (14, 1), (7, 4)])
def test_roi_align_bwd_pipe_bit_identical(P, n_rot):
    """The chunk-pipelined backward (LOFT_ROI_BWD_PIPE: global -> LDS copies one chunk ahead, one barrier per chunk, tables two
    pairs ahead) runs the shipped serial kernel's arithmetic in the serial kernel's order: the maps must be torch.equal -- small RoIs
    (many chunks per pair at P = 14), RoIs without any weight in a tile they touch, > 256 RoIs per (image, tile) batch,
    the accumulate mode."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(5 + P + n_rot)
    B, C, size = 2, 256, 256
    strides = [4, 8, 16, 32]
    shapes = [(B, C, size // s, size // s) for s in strides]
    for n, wmin, wmax in ((300, 4, 400), (1500, 8, 60), (700, 3, 20)):
        rois = _rand_rois(rng, n, B, size, wmin, wmax)
        if n == 700:
            rois[:400, 1:] = torch.tensor([40., 40., 52., 52.]) + torch.rand(400, 4)      # > 256 RoIs on one tile of image 0 / 1
        rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().cuda()
        g = torch.randn(n_rot * rois.shape[0], C, P, P, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        outs = {}
        for name, var in (('pipe', K.ROI_BWD_PIPE), ('serial', K.ROI_AUTO)):
            K.ROI_BWD_VARIANT = var
            try:
                first = K.roi_align_bwd(g, rois, shapes, P, strides, n_rot=n_rot, rois_sorted=True, out_dtype=torch.bfloat16)
                twice = K.roi_align_bwd(g, rois, shapes, P, strides, n_rot=n_rot, rois_sorted=True,
                                        grad_feats=[x.clone() for x in first])
                unsorted = K.roi_align_bwd(g, rois, shapes, P, strides, n_rot=n_rot, rois_sorted=False, out_dtype=torch.bfloat16)
            finally:
                K.ROI_BWD_VARIANT = K.ROI_AUTO
            outs[name] = (first, twice, unsorted)
        for a, b in zip(outs['pipe'], outs['serial']):
            for x, y in zip(a, b):
                assert torch.equal(x, y), (n, P, n_rot)
        assert any(x.float().abs().sum().item() > 0 for x in outs['pipe'][0])


def test_roi_align_variants_selected_in_process():
    """include/loft_hip.h LOFT_ROI_*: the explicit kernel selectors of loft_roi_align_{fwd,bwd}_v (no environment variables):
    the sample-order forward against the separable one, the tile-owner VALU backward against the per-pair GEMMs, 16-bit maps."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(11)
    B, C, size, P = 2, 256, 256, 7
    strides = [4, 8, 16, 32]
    rois = _rand_rois(rng, 300, B, size)
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().cuda()
    torch.manual_seed(3)
    feats = [torch.randn(B, C, size // s, size // s, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
             for s in strides]
    g = torch.randn(rois.shape[0], C, P, P, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    shapes = [tuple(f.shape) for f in feats]
    try:
        sep = K.roi_align_fwd(feats, rois, P, strides)
        mfma = K.roi_align_bwd(g, rois, shapes, P, strides, rois_sorted=True, out_dtype=torch.bfloat16)
        K.ROI_FWD_VARIANT, K.ROI_BWD_VARIANT = K.ROI_FWD_SAMPLE, K.ROI_BWD_VALU
        smp = K.roi_align_fwd(feats, rois, P, strides)
        valu = K.roi_align_bwd(g, rois, shapes, P, strides, rois_sorted=True, out_dtype=torch.bfloat16)
        K.ROI_FWD_VARIANT = K.ROI_FWD_SEP4
        sep4 = K.roi_align_fwd(feats, rois, P, strides)
    finally:
        K.ROI_FWD_VARIANT = K.ROI_BWD_VARIANT = K.ROI_AUTO
    for s_ in (sep, sep4):          # the 16-byte (shipped) and the 8-byte separable forwards against the sample-order kernel
        assert (s_.float() - smp.float()).abs().max().item() <= 2 ** -6 * max(1.0, smp.float().abs().max().item())
        assert (s_.float() - smp.float()).abs().mean().item() <= 2e-3
    for a, b in zip(mfma, valu):
        scale = max(1.0, b.float().abs().max().item())
        assert (a.float() - b.float()).abs().max().item() <= 8e-3 * scale
    K.ROI_FWD_VARIANT = 7
    try:
        with pytest.raises(K.L.LoftHipError):
            K.roi_align_fwd(feats, rois, P, strides)
    finally:
        K.ROI_FWD_VARIANT = K.ROI_AUTO


def test_roi_align_bwd_multi_matches_sum_of_lists():
    """loft_roi_align_bwd_multi: the three extractors' lists (7x7; 14x14; 14x14 x 4 rotations) in one pass per level ==
    the fp32 maps of the three lists added up (each pinned to the oracle above), incl. an empty list, the accumulate mode,
    and the list-after-list route the library takes for C != 256."""
    from bonai_amd import kernels as K
    rng = np.random.RandomState(23)
    B, size = 2, 256
    strides = [4, 8, 16, 32]
    for C in (256, 64):
        shapes = [(B, C, size // s, size // s) for s in strides]
        sets = []
        for n, P, n_rot in ((500, 7, 1), (90, 14, 1), (0, 14, 1), (60, 14, 4)):
            rois = _rand_rois(rng, n, B, size) if n else torch.zeros(0, 5)
            rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().cuda()
            g = torch.randn(n_rot * n, C, P, P, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
            sets.append((g, rois, P, n_rot, True))
        want = [torch.zeros(s[0], s[2], s[3], s[1], device='cuda').permute(0, 3, 1, 2) for s in shapes]
        for g, rois, P, n_rot, _ in sets:
            if rois.shape[0]:
                for w, x in zip(want, K.roi_align_bwd(g, rois, shapes, P, strides, n_rot=n_rot, rois_sorted=True,
                                                      out_dtype=torch.float32)):
                    w += x
        for group in (sets[:3], [sets[0], sets[1], sets[3]], [sets[2]]):
            ref = [torch.zeros_like(w) for w in want]
            for g, rois, P, n_rot, _ in group:
                if rois.shape[0]:
                    for w, x in zip(ref, K.roi_align_bwd(g, rois, shapes, P, strides, n_rot=n_rot, rois_sorted=True,
                                                          out_dtype=torch.float32)):
                        w += x
            got = K.roi_align_bwd_multi(group, shapes, strides)
            for gg, ww in zip(got, ref):
                assert gg.dtype == torch.bfloat16 and gg.shape == ww.shape
                scale = max(1.0, ww.abs().max().item())
                tol = 6e-3 if C == 256 else 2e-2          # one rounding fused; one per list on the list-after-list route
                assert (gg.float() - ww).abs().max().item() <= tol * scale
            twice = K.roi_align_bwd_multi(group, shapes, strides, grad_feats=[x.clone() for x in got])
            for tt, ww in zip(twice, ref):
                scale = max(1.0, ww.abs().max().item())
                assert (tt.float() - 2 * ww).abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize('K_,B,P,n_rot', [(8192, 8, 7, 1), (1001, 3, 14, 1), (777, 2, 7, 4), (256, 1, 7, 1)])
def test_roi_align_launch_order_is_a_permutation_and_changes_nothing(K_, B, P, n_rot):
    """loft_roi_order + loft_roi_align_fwd_ord: the (image, level, row strip) launch order is a permutation
    of the list whose buckets ascend, and the ordered launch writes bit-identical features (same arithmetic, same output rows) --
    for K not a multiple of 8, for the four FOA rotations, for fp32 maps as well."""
    import ctypes
    from bonai_amd import kernels as K, lib as L
    from bonai_amd.debug import DBG
    rng = np.random.RandomState(K_)
    C, size, strides = 64, 512, [4, 8, 16, 32]
    torch.manual_seed(1)
    rois = _rand_rois(rng, K_, B, size).cuda()
    for dtype in (torch.bfloat16, torch.float32):
        feats = [torch.randn(B, C, size // s, size // s, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
                 for s in strides]
        plain = K.roi_align_fwd(feats, rois, P, strides, n_rot=n_rot)          # (shipped: list order)
        prev, K.ROI_FWD_SORT_MIN = K.ROI_FWD_SORT_MIN, 256
        try:
            ordered = K.roi_align_fwd(feats, rois, P, strides, n_rot=n_rot)
            with DBG.override(no_roi_sort=True):
                assert torch.equal(plain, K.roi_align_fwd(feats, rois, P, strides, n_rot=n_rot))
        finally:
            K.ROI_FWD_SORT_MIN = prev
        assert torch.equal(plain, ordered), dtype
    H, W, S = K._level_args(feats, strides)
    order = torch.full((K_,), -1, dtype=torch.int32, device='cuda')
    L.check(L.load().loft_roi_order(H, S, 4, 56, L.ptr(rois), K_, B, L.ptr(order), L.stream()), 'loft_roi_order')
    o = order.long().cpu()
    assert torch.equal(o.sort()[0], torch.arange(K_))
    r = rois.cpu()[o]
    lv = ops_ref.map_roi_levels(r)
    nstrip = min(32, 2048 // (4 * B))
    hs = torch.tensor([size // s for s in strides], dtype=torch.float32)
    cy = 0.5 * (r[:, 2] + r[:, 4]) / torch.tensor(strides, dtype=torch.float32)[lv]
    strip = (cy * nstrip / hs[lv]).floor().clamp(0, nstrip - 1).long()   # (float32 like the kernel: an RoI ON a strip edge may differ)
    key = (r[:, 0].long() * 4 + lv) * nstrip + strip
    assert int((key[1:] < key[:-1]).sum()) <= max(2, K_ // 500), 'buckets must ascend along the launch order'


def test_rpn_level_fused_launches_match_the_per_level_ones():
    """loft_rpn_scores_levels / loft_rpn_decode_levels / loft_nms_segmented_levels / loft_rpn_finalize (the training step's proposal
    chain in seven launches) against the per-level launches and tensor expressions they replaced: scores, candidate boxes and
    scores, the per-image coordinate maximum, the keep flags under the device-derived level shift and the final proposal table --
    all torch.equal."""
    from bonai_amd import kernels as K
    torch.manual_seed(3)
    B, A = 3, 3
    sizes = [(20, 24), (10, 12), (5, 6)]
    strides = [4, 8, 16]
    heads = [torch.randn(B, 16, h, w).cuda().contiguous(memory_format=torch.channels_last) for h, w in sizes]
    n_l = [h * w * A for h, w in sizes]
    lvl_off = [0, n_l[0], n_l[0] + n_l[1], sum(n_l)]
    N = lvl_off[-1]
    keys1 = torch.empty(B * N, device='cuda')
    for f, o in zip(heads, lvl_off):
        K.rpn_scores(f, A, N, o, keys1)
    keys2 = torch.empty(B * N, device='cuda')
    img_max = torch.full((B,), 7.0, device='cuda')
    K.rpn_scores_levels(heads, A, N, lvl_off, keys2, img_max)
    assert torch.equal(keys1, keys2) and torch.isinf(img_max).all() and (img_max < 0).all()
    seg = torch.tensor([b * N + o for b in range(B) for o in lvl_off[:-1]] + [B * N], dtype=torch.int64, device='cuda')
    topk = [min(200, n) for n in n_l]
    skeys, sidx = K.segmented_sort_desc(keys1, seg)
    coff = [0, topk[0], topk[0] + topk[1], sum(topk)]
    C = coff[-1]
    base = [torch.tensor([[-s * 2., -s, s * 2., s], [-s * 1.5, -s * 1.5, s * 1.5, s * 1.5], [-s, -s * 2., s, s * 2.]], device='cuda') * 4 for s in strides]
    means, stds, shape = (0., 0., 0., 0.), (1., 1., 1., 1.), (80, 96)
    cand1 = torch.empty(B, C, 4, device='cuda')
    for l, f in enumerate(heads):
        K.rpn_decode(f, sidx, A, N, lvl_off[l], topk[l], base[l], strides[l], means, stds, shape, C, coff[l], cand1)
    cand2 = torch.empty(B, C, 4, device='cuda')
    cs2 = torch.empty(B, C, device='cuda')
    K.rpn_decode_levels(heads, sidx, skeys, A, N, lvl_off, topk, base, strides, means, stds, shape, C, coff, cand2, cs2, img_max)
    sk = skeys.view(B, N)
    cs1 = torch.cat([sk[:, lvl_off[l]:lvl_off[l] + topk[l]] for l in range(3)], 1)
    assert torch.equal(cand1, cand2) and torch.equal(cs1, cs2)
    assert torch.equal(img_max, cand1.view(B, -1).amax(1))
    nseg = torch.tensor([b * C + o for b in range(B) for o in coff[:-1]] + [B * C], dtype=torch.int64, device='cuda')
    shift = (torch.arange(3, device='cuda', dtype=torch.float32)[None] * (img_max[:, None] + 1)).reshape(-1)
    keep1 = K.nms_segmented(cand1.view(-1, 4), nseg, 0.7, seg_shift=shift, max_segment=max(topk))
    keep2 = K.nms_segmented(cand1.view(-1, 4), nseg, 0.7, max_segment=max(topk), img_max=img_max, levels=3, covered=True)
    assert torch.equal(keep1, keep2) and 0 < int(keep1.sum()) < keep1.numel()
    post = 150
    iseg = torch.arange(B + 1, dtype=torch.int64, device='cuda') * C
    masked = torch.where(keep1.view(B, C).bool(), cs1, -1.0)
    fs, fi = K.segmented_sort_desc(masked.reshape(-1), iseg)
    fs = fs.view(B, C)[:, :post]; fi = fi.view(B, C)[:, :post].long()
    valid = fs >= 0
    want = torch.cat([cand1.view(-1, 4)[fi.reshape(-1)].view(B, post, 4), fs[..., None]], -1) * valid[..., None]
    ts, ti = K.segmented_topk_desc(cs1.reshape(-1), iseg, post, max_segment=C, key_mask=keep1)
    props, counts = K.rpn_finalize(ts, ti, cand1.view(-1, 4), B, C, post)
    assert torch.equal(props, want) and torch.equal(counts, valid.sum(1))
