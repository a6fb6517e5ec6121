"""ctypes front-end of the plain-C oracle (oracle/loft_oracle.c).  TEST INFRASTRUCTURE.

Array convention: torch CPU tensors, fp32 / int64, the reference's NCHW layout.
"""
import ctypes

import torch

from . import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_oracle.build())
        _lib.orc_nms.restype = ctypes.c_int64
        _lib.orc_nms_pred.restype = ctypes.c_int64
        _lib.orc_soft_nms.restype = ctypes.c_int64
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous().cpu()


def roi_align_fwd(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """feat [N,C,H,W], rois [K,5] (batch_idx, x1, y1, x2, y2) -> [K,C,ph,pw]."""
    feat, rois = _f32(feat), _f32(rois)
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    N, C, H, W = feat.shape
    K = rois.shape[0]
    out = torch.zeros(K, C, ph, pw, dtype=torch.float32)
    if K:
        lib().orc_roi_align_fwd(_p(feat), N, C, H, W, _p(rois), K, ph, pw, ctypes.c_float(spatial_scale),
                                int(sampling_ratio), int(aligned), _p(out))
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale, sampling_ratio=0, aligned=True):
    grad_out, rois = _f32(grad_out), _f32(rois)
    N, C, H, W = feat_shape
    K, _, ph, pw = grad_out.shape
    gin = torch.zeros(N, C, H, W, dtype=torch.float32)
    if K:
        lib().orc_roi_align_bwd(_p(grad_out), N, C, H, W, _p(rois), K, ph, pw, ctypes.c_float(spatial_scale),
                                int(sampling_ratio), int(aligned), _p(gin))
    return gin


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, spatial_scale, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat.shape), out_size, spatial_scale, sampling_ratio, aligned)
        return roi_align_fwd(feat, rois, out_size, spatial_scale, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, _, scale, sr, al = ctx.meta
        return roi_align_bwd(g, rois, shape, scale, sr, al), None, None, None, None, None


def roi_align(feat, rois, out_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
    """Differentiable (torch-autograd) oracle RoIAlign, mmcv.ops.roi_align argument order."""
    assert pool_mode == 'avg'
    return _RoIAlignFn.apply(feat, rois, out_size, float(spatial_scale), int(sampling_ratio), bool(aligned))


def argsort_desc(scores):
    scores = _f32(scores)
    order = torch.empty(scores.numel(), dtype=torch.int64)
    lib().orc_argsort_desc(_p(scores), ctypes.c_int64(scores.numel()), _p(order))
    return order


def nms(boxes, scores, iou_threshold, predicate='device'):
    """-> (dets [M,5], keep [M] int64), keep in score-descending order (mmcv.ops.nms contract).
    predicate: 'device' = mmcv-1.0.5's CUDA kernel (inter > thr*union), 'cpu' = its host nms_cpu (inter/union >= thr)."""
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    keep = torch.empty(n, dtype=torch.int64)
    pred = {'device': 0, 'cpu': 1}[predicate]
    nk = lib().orc_nms_pred(_p(boxes), _p(scores), ctypes.c_int64(n), ctypes.c_float(iou_threshold), pred, _p(keep)) if n else 0
    keep = keep[:nk]
    dets = torch.cat([boxes[keep], scores[keep, None]], dim=1)
    return dets, keep


def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear'):
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    dets = torch.zeros(n, 5, dtype=torch.float32)
    inds = torch.zeros(n, dtype=torch.int64)
    m = {'naive': 0, 'linear': 1, 'gaussian': 2}[method]
    nk = lib().orc_soft_nms(_p(boxes), _p(scores), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                            ctypes.c_float(sigma), ctypes.c_float(min_score), m, _p(dets), _p(inds)) if n else 0
    return dets[:nk], inds[:nk]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv-1.0.5 batched_nms: shift boxes by idx*(max_coord+1), run the op named in nms_cfg."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    boxes = _f32(boxes)
    scores = _f32(scores)
    if class_agnostic or boxes.numel() == 0:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)
        boxes_for_nms = boxes + offsets[:, None]
    op = {'nms': nms, 'soft_nms': soft_nms}[cfg.pop('type', 'nms')]
    dets, keep = op(boxes_for_nms, scores, **cfg)
    return torch.cat([boxes[keep], dets[:, -1:]], dim=-1), keep
