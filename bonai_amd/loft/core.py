"""Host-side mirrors of the reference's core box machinery, executing on the HIP kernels:
AnchorGenerator (mmdet/core/anchor/anchor_generator.py:9-329), DeltaXYWHBBoxCoder
(core/bbox/coder/delta_xywh_bbox_coder.py:9-197), DeltaXYOffsetCoder (delta_xy_offset_coder.py:19-88),
MaxIoUAssigner (core/bbox/assigners/max_iou_assigner.py:9-212), RandomSampler
(core/bbox/samplers/random_sampler.py:8-75 + base_sampler.py:34-101), BboxOverlaps2D.

Everything here is *batched over the images of the step* and stays on the device; the reference's
per-image python loops and its per-gt low-quality loop are gone (SURVEY.md section 7 "hard parts" (d)).
"""
import numpy as np
import torch

from .. import kernels as K
from .builder import ANCHOR_GENERATORS, BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS


@ANCHOR_GENERATORS.register_module()
class AnchorGenerator:
    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if scales is None or octave_base_scale is not None or centers is not None or center_offset != 0. or not scale_major:
            raise NotImplementedError('only the (scales, ratios, strides) form used by configs/loft_foa')
        # anchor_generator.py:72-76: a stride is an int or an (x, y) pair; base size = the smaller of the pair
        self.stride_pairs = [(int(s), int(s)) if not isinstance(s, (tuple, list)) else (int(s[0]), int(s[1])) for s in strides]
        self.strides = [sx if sx == sy else (sx, sy) for sx, sy in self.stride_pairs]
        self.base_sizes = [min(p) for p in self.stride_pairs] if base_sizes is None else list(base_sizes)
        if len(self.base_sizes) != len(self.strides):
            raise ValueError(f'The number of strides should be the same as base sizes, got {self.strides} and {self.base_sizes}')
        self.scales = torch.tensor(scales, dtype=torch.float32)
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self.base_anchors = [self._base(b) for b in self.base_sizes]
        self._cache = {}

    def _base(self, base_size):
        # anchor_generator.py:142-181, same fp32 op order so anchors are bit-identical
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        ws = (base_size * w_ratios[:, None] * self.scales[None, :]).view(-1)
        hs = (base_size * h_ratios[:, None] * self.scales[None, :]).view(-1)
        return torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], dim=-1)

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    def grid_anchors(self, featmap_sizes, device='cuda'):
        key = (tuple(tuple(int(v) for v in s) for s in featmap_sizes), str(device))
        if key not in self._cache:
            out = []
            if len(featmap_sizes) != self.num_levels:
                raise ValueError(f'{len(featmap_sizes)} feature maps for {self.num_levels} anchor levels')
            for (h, w), (stx, sty), ba in zip(featmap_sizes, self.stride_pairs, self.base_anchors):
                sx = torch.arange(0, int(w), dtype=torch.float32) * stx
                sy = torch.arange(0, int(h), dtype=torch.float32) * sty
                xx = sx.repeat(int(h))
                yy = sy.view(-1, 1).repeat(1, int(w)).view(-1)
                shifts = torch.stack([xx, yy, xx, yy], dim=-1)
                out.append((ba[None] + shifts[:, None]).view(-1, 4).to(device))
            if torch.device(device).type == 'cuda':      # cached across streams: complete before anyone reads it (rpn._static_ready)
                torch.cuda.current_stream().synchronize()
            self._cache[key] = out
        return self._cache[key]


class _Coder:
    def __init__(self, target_means, target_stds):
        self.means = tuple(float(v) for v in target_means)
        self.stds = tuple(float(v) for v in target_stds)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder(_Coder):
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.)):
        super().__init__(target_means, target_stds)

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        return K.bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0) and pred_bboxes.size(1) == 4
        return K.delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip)


@BBOX_CODERS.register_module()
class DeltaXYOffsetCoder(_Coder):
    def __init__(self, target_means=(0., 0.), target_stds=(0.5, 0.5)):
        super().__init__(target_means, target_stds)

    def encode(self, bboxes, gt_offsets):
        """delta_xy_offset_coder.py:28-32 -> offset2delta :46-65."""
        return K.offset_targets(bboxes, gt_offsets, self.means, self.stds, 2)

    def decode(self, bboxes, pred_offsets, max_shape=None, wh_ratio_clip=16 / 1000):
        """delta_xy_offset_coder.py:34-43 -> delta2offset :67-88 (no clamp when max_shape is None)."""
        return K.offset_decode(pred_offsets, bboxes, self.means, self.stds, max_shape if max_shape is not None else (3.0e38, 3.0e38))


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    pass


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        if not gt_max_assign_all or ignore_iof_thr > 0 or isinstance(neg_iou_thr, tuple):
            raise NotImplementedError('assigner variant not used by configs/loft_foa')
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = float(pos_iou_thr), float(neg_iou_thr), float(min_pos_iou)
        self.match_low_quality = match_low_quality

    def assign_batched(self, boxes, nbox, gts, ngt):
        """boxes [B,N,4], nbox [B], gts [B,K,4], ngt [B] (device) -> (gt_inds int64 [B,N], max_overlaps [B,N])."""
        return K.iou_assign(boxes, nbox, gts, ngt, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou,
                            self.match_low_quality)


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """choice_mode 'random' = uniform without replacement (random_sampler.py:31-55);
    'first' = lowest indices, the deterministic rule parity runs inject on both sides."""
    choice_mode = 'random'

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        if neg_pos_ub >= 0:
            raise NotImplementedError('neg_pos_ub')
        self.num, self.pos_fraction, self.add_gt_as_proposals = int(num), float(pos_fraction), add_gt_as_proposals

    def sample_batched(self, gt_inds):
        """gt_inds int64 [B,N].  -> dict(pos_idx [B,P], pos_valid [B,P], neg_idx [B,Q], neg_valid [B,Q]) with the
        chosen indices in ascending order inside each image (base_sampler.py:86,96 `.unique()`)."""
        B, N = gt_inds.shape
        dev = gt_inds.device
        P = min(int(self.num * self.pos_fraction), N)
        Q = min(self.num, N)
        if gt_inds.is_cuda and N > 0:       # the product path: one launch (radix select on hashed keys + ordered compaction)
            pidx, pvalid, nidx, nvalid = K.random_sample(gt_inds, self.num, int(self.num * self.pos_fraction), self.choice_mode)
            return dict(pos_idx=pidx, pos_valid=pvalid, neg_idx=nidx, neg_valid=nvalid)
        return self.sample_batched_host(gt_inds)

    def sample_batched_host(self, gt_inds):
        """The same contract with torch ops: host tensors in the CPU-side logic tests, and the checker of the kernel."""
        B, N = gt_inds.shape
        dev = gt_inds.device
        P = min(int(self.num * self.pos_fraction), N)
        Q = min(self.num, N)
        if self.choice_mode == 'first':
            key = torch.arange(N, dtype=torch.float32, device=dev).expand(B, N)
        else:
            key = torch.rand(B, N, device=dev)
        inf = torch.full((), float('inf'), device=dev)
        pv, pidx = torch.topk(torch.where(gt_inds > 0, key, inf), P, dim=1, largest=False)
        pvalid = pv < inf
        npos = pvalid.sum(1)
        nv, nidx = torch.topk(torch.where(gt_inds == 0, key, inf), Q, dim=1, largest=False)
        quota = (self.num - npos)[:, None]
        nvalid = (nv < inf) & (torch.arange(Q, device=dev)[None] < quota)
        big = torch.full((), N, device=dev, dtype=pidx.dtype)
        pidx, order = torch.sort(torch.where(pvalid, pidx, big), dim=1)
        pvalid = torch.gather(pvalid, 1, order)
        nidx, order = torch.sort(torch.where(nvalid, nidx, big), dim=1)
        nvalid = torch.gather(nvalid, 1, order)
        return dict(pos_idx=pidx.clamp(max=N - 1), pos_valid=pvalid, neg_idx=nidx.clamp(max=N - 1), neg_valid=nvalid)


def pad_rows(rows, device, dtype, kmax=None):
    """list[[K_i, ...]] -> [B, Kmax, ...] zero-padded.  One kernel when every K_i equals Kmax (stack); otherwise one cat and
    one indexed copy -- not one copy kernel per image."""
    B = len(rows)
    ks = [int(r.shape[0]) for r in rows]
    kmax = kmax or max(1, max(ks))
    rows = [r.to(device=device, dtype=dtype) for r in rows]
    if all(k == kmax for k in ks):
        return torch.stack(rows)
    out = torch.zeros((B, kmax) + tuple(rows[0].shape[1:]), dtype=dtype, device=device)
    if sum(ks):
        idx = K.h2d([i * kmax + j for i, k in enumerate(ks) for j in range(k)], torch.int64, device)
        out.flatten(0, 1).index_copy_(0, idx, torch.cat(rows))
    return out


_PAD_GTS_CACHE = [None, None, None]      # the RPN head and the RoI head pad the same list in the same step


def pad_gts(gt_bboxes, device):
    """list[[K_i,4]] -> (gts [B,Kmax,4] fp32, ngt int32 [B])."""
    key = (str(device),) + tuple((g.data_ptr(), g._version, tuple(g.shape)) for g in gt_bboxes)
    if _PAD_GTS_CACHE[0] == key:
        return _PAD_GTS_CACHE[2]
    gts = pad_rows(gt_bboxes, device, torch.float32)
    ngt = K.h2d([int(g.shape[0]) for g in gt_bboxes], torch.int32, device)
    _PAD_GTS_CACHE[:] = [key, list(gt_bboxes), (gts, ngt)]      # (the tensors are kept alive: their addresses are the key)
    # the assigner's result for the gt boxes the RoI sampler adds to the proposals (add_gt_as_proposals): box j of an image is
    # assigned to gt j + 1, padding -1.  Built here, with the padded boxes -- i.e. before the backbone when the RPN head prefetches
    # its targets -- not as five small launches between the proposals and the RoI heads.
    ar = torch.arange(gts.shape[1], device=gts.device)[None]
    _GT_SELF[0] = torch.where(ar < ngt[:, None], ar + 1, torch.full_like(ar, -1)).expand(len(gt_bboxes), -1)
    return gts, ngt


_GT_SELF = [None]


def gt_self_inds(gt_bboxes, device):
    """int64 [B, Kmax]: j + 1 for the valid gt slots of pad_gts' layout, -1 for padding."""
    pad_gts(gt_bboxes, device)
    return _GT_SELF[0]
