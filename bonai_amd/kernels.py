"""torch-tensor front-ends of the C-ABI kernels (include/loft_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream and autograd bookkeeping.
Every function launches hand-written gfx950 kernels through ctypes; nothing falls back to
eager PyTorch or the CPU (bonai_amd.lib raises instead).

Layout contract: 4-D activations are NCHW-*shaped* tensors with channels_last strides, i.e. NHWC
in memory (SURVEY.md section 8b "Tensor conventions"), so reference checkpoints and call sites
keep their shapes while the kernels see channel-contiguous pixels.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as L
from .debug import DBG as _DBG     # A/B switches (bonai_amd/debug.py)

c_int, c_int64, c_float, c_void_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


def h2d(values, dtype, device):
    """Small host list -> device tensor WITHOUT stalling the host: a pageable-memory H2D copy is stream-ordered and blocks the
    launching thread until every kernel queued before it has run (an implicit device sync per call -- six per step before);
    from pinned memory (PyTorch's caching pinned allocator) it is asynchronous."""
    t = torch.from_numpy(values).to(dtype) if isinstance(values, np.ndarray) else torch.tensor(values, dtype=dtype)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    if H2D_KEEP is not None:
        # hipGraph capture (bonai_amd/graphs.py): a table uploaded inside a captured section holds addresses that are static
        # for the graph's lifetime, so it is uploaded ONCE, now, on a stream that is not capturing (no memcpy node, no pinned
        # buffer the replay would read again), and kept alive by the graph's owner.
        # (torch hands out streams from a pool of 32 per device, round robin: a stream drawn long ago may be the SAME hardware queue
        #  as the capture stream or one of its forks drawn now -- the upload would then be captured and the synchronize below is
        #  illegal; seen in round 6 as "capturing stream has unjoined work" in one test order of the suite.  Draw until the stream
        #  is not part of a capture.)
        global _H2D_STREAM
        for _ in range(64):
            if _H2D_STREAM is None:
                _H2D_STREAM = torch.cuda.Stream()
            with torch.cuda.stream(_H2D_STREAM):
                if not torch.cuda.is_current_stream_capturing():
                    break
            _H2D_STREAM = None
        else:
            raise L.LoftHipError('no stream outside the running capture for a descriptor upload')
        with torch.cuda.stream(_H2D_STREAM):
            d = t.to(device)
        _H2D_STREAM.synchronize()
        H2D_KEEP.append(d)
        return d
    return t.pin_memory().to(device, non_blocking=True)


H2D_KEEP = None       # a list while bonai_amd.graphs.FeatureGraphs.capture() records a section
_H2D_STREAM = None


class _ZeroPool:
    """One zeroed fp32 slab per training step for the many small accumulation buffers of the backward (every weight-gradient
    launch accumulates into a zeroed [taps, Cout, Cin] buffer with atomics: ~190 separate fill launches per step otherwise).
    The trainer calls zero_pool_begin() at the start of a step: ONE memset of the slab; pooled_zeros() then hands out views.
    The slab is sized from the previous step's requests; anything that does not fit falls back to torch.zeros."""
    buf = None
    off = 0
    need = 0
    active = False


def zero_pool_begin(device):
    zp = _ZeroPool
    if zp.buf is None or zp.need > zp.buf.numel() or zp.buf.device != torch.device(device):
        if zp.need > 0:
            zp.buf = torch.empty(int(zp.need * 1.05) + 1024, dtype=torch.float32, device=device)
    if zp.buf is not None:
        zp.buf.zero_()
    zp.off, zp.need, zp.active = 0, 0, zp.buf is not None
    sp = _ScratchPool
    if sp.need > 0 and (sp.buf is None or sp.need > sp.buf.numel() or sp.buf.device != torch.device(device)):
        sp.buf = None                                   # (release before the larger request)
        sp.buf = torch.empty(int(sp.need * 1.05) + 1024, dtype=torch.float32, device=device)
    sp.off, sp.need, sp.active = 0, 0, sp.buf is not None


def zero_pool_end():
    _ZeroPool.active = False
    _ScratchPool.active = False


class _ScratchPool:
    """The same per-step slab for buffers that need NO initialisation (the split-K slots of the weight gradients: ~3 GB per step
    at batch 8 x 1024^2, each written completely by its launch and read once by the batched unpack).  Bump allocation, reset
    with the zero pool at the start of a step; without it every slot buffer is a caching-allocator block that several streams
    touch (record_stream delays its reuse), and the allocator keeps growing and trimming."""
    buf = None
    off = 0
    need = 0
    active = False


def pooled_scratch(shape, device):
    """Uninitialised fp32 of `shape`: a view of the step's scratch slab when the trainer opened one, else torch.empty."""
    n = 1
    for d in shape:
        n *= int(d)
    n64 = (n + 63) // 64 * 64
    sp = _ScratchPool
    sp.need += n64
    if sp.active and sp.buf.device == torch.device(device) and sp.off + n64 <= sp.buf.numel():
        v = sp.buf[sp.off:sp.off + n].view(*shape)
        sp.off += n64
        return v
    return torch.empty(*shape, dtype=torch.float32, device=device)


def pooled_zeros(shape, device):
    """fp32 zeros of `shape`: a view of the step's pre-zeroed slab when the trainer opened one, else torch.zeros."""
    n = 1
    for d in shape:
        n *= int(d)
    n64 = (n + 63) // 64 * 64
    zp = _ZeroPool
    zp.need += n64
    if zp.active and zp.buf.device == torch.device(device) and zp.off + n64 <= zp.buf.numel():
        v = zp.buf[zp.off:zp.off + n].view(*shape)
        zp.off += n64
        return v
    return torch.zeros(*shape, dtype=torch.float32, device=device)


def _nhwc(t):
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise L.LoftHipError('expected a 4-D channels_last (NHWC-in-memory) tensor, got strides '
                             f'{tuple(t.stride())} for shape {tuple(t.shape)}')
    return t


def empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def zeros_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last).zero_()


# ------------------------------------------------------------------ RoIAlign

def _level_args(feats, strides):
    H = L.arr(c_int, [f.shape[2] for f in feats])
    W = L.arr(c_int, [f.shape[3] for f in feats])
    S = L.arr(c_float, [1.0 / s for s in strides])
    return H, W, S


# include/loft_hip.h LOFT_ROI_*: kernel selector of loft_roi_align_{fwd,bwd}_v (0 = the shipped choice); tests set these
ROI_AUTO, ROI_FWD_SAMPLE, ROI_FWD_SEP4, ROI_BWD_VALU, ROI_BWD_PIPE = 0, 1, 2, 1, 2
ROI_FWD_VARIANT = ROI_AUTO
# RoI lists at least this long are launched in (image, level, row strip) order (loft_roi_order); None = list order (shipped).
# Measured round 3 (rocprofv3, bench step at 256 positives / image, same box): roi_align_fwd_sep_kernel 243.7 us per launch
# ordered against 248.3 us in list order (-2 %), plus 14 us for the ordering launch in front of it; step 37.70-37.96 ms ordered
# against 37.61-37.82 ms: the forward is bound by its per-RoI chain of dependent L2 round trips, not by where the windows come
# from, so fetching them from HBM once per XCD instead of once per RoI buys nothing.  Kept selectable and tested, not shipped.
ROI_FWD_SORT_MIN = None
ROI_BWD_VARIANT = ROI_AUTO


def roi_align_fwd(feats, rois, P, strides, finest_scale=56, n_rot=1):
    lib = L.load()
    L.dev_check(rois, *feats)
    feats = [_nhwc(f) for f in feats]
    rois = rois.float().contiguous()
    K, C = rois.shape[0], feats[0].shape[1]
    out = empty_nhwc(n_rot * K, C, P, P, feats[0].dtype, rois.device)
    if K == 0:
        return out
    H, W, S = _level_args(feats, strides)
    fp = L.arr(c_void_p, [f.data_ptr() for f in feats])
    order = None
    sort_min = ROI_FWD_SORT_MIN if ROI_FWD_SORT_MIN is not None else (256 if _DBG.roi_sort else None)
    if sort_min is not None and K >= sort_min and not _DBG.no_roi_sort:
        # launch order (image, level, row strip): one XCD walks one contiguous eighth of it, overlapping windows meet in its L2
        order = torch.empty(K, dtype=torch.int32, device=rois.device)
        L.check(lib.loft_roi_order(H, S, len(feats), int(finest_scale), L.ptr(rois), K, int(feats[0].shape[0]), L.ptr(order),
                                   L.stream()), 'loft_roi_order')
    L.check(lib.loft_roi_align_fwd_ord(fp, H, W, S, len(feats), int(finest_scale), C, L.dtype_code(feats[0]), L.ptr(rois),
                                       K, int(P), int(n_rot), L.ptr(out), int(ROI_FWD_VARIANT), L.ptr(order), L.stream()),
            'loft_roi_align_fwd_ord')
    return out


def roi_align_bwd(grad_out, rois, feat_shapes, P, strides, finest_scale=56, n_rot=1, grad_feats=None,
                  rois_sorted=False, out_dtype=torch.float32):
    """Writes (or, with grad_feats supplied, accumulates into) NHWC gradient maps of out_dtype and returns them."""
    lib = L.load()
    L.dev_check(grad_out, rois)
    grad_out = _nhwc(grad_out)
    rois = rois.float().contiguous()
    K, C = rois.shape[0], grad_out.shape[1]
    accumulate = grad_feats is not None
    if grad_feats is None:
        mk = zeros_nhwc if K == 0 else empty_nhwc
        grad_feats = [mk(s[0], s[1], s[2], s[3], out_dtype, rois.device) for s in feat_shapes]
    if K == 0:
        return grad_feats
    H = L.arr(c_int, [s[2] for s in feat_shapes])
    W = L.arr(c_int, [s[3] for s in feat_shapes])
    S = L.arr(c_float, [1.0 / s for s in strides])
    gp = L.arr(c_void_p, [g.data_ptr() for g in grad_feats])
    ws = torch.empty(48 * K, dtype=torch.uint8, device=rois.device)
    L.check(lib.loft_roi_align_bwd_v(gp, H, W, S, len(grad_feats), int(finest_scale), C, L.dtype_code(grad_out),
                                     L.ptr(rois), K, int(P), int(n_rot), L.ptr(grad_out), int(feat_shapes[0][0]),
                                     int(accumulate), int(rois_sorted), L.ptr(ws), L.dtype_code(grad_feats[0]),
                                     int(ROI_BWD_VARIANT), L.stream()), 'loft_roi_align_bwd_v')
    return grad_feats


def roi_align_bwd_multi(sets, feat_shapes, strides, finest_scale=56, grad_feats=None, out_dtype=None, out=None):
    """Backward of several RoIAlign calls over the same pyramid, each map pixel written once (loft_roi_align_bwd_multi).

    sets: [(grad_out, rois, P, n_rot, rois_sorted), ...]; writes (or with grad_feats accumulates into) NHWC maps."""
    lib = L.load()
    out_dtype = out_dtype or L.act16()
    dev = sets[0][0].device
    accumulate = grad_feats is not None
    if grad_feats is None:      # (out: preallocated maps to write instead of fresh ones)
        grad_feats = out if out is not None else [empty_nhwc(s[0], s[1], s[2], s[3], out_dtype, dev) for s in feat_shapes]
    gos = [_nhwc(s[0]) for s in sets]
    rois = [s[1].float().contiguous() for s in sets]
    L.dev_check(*gos, *rois)
    Ks = [r.shape[0] for r in rois]
    wss = [torch.empty(max(48 * k, 48), dtype=torch.uint8, device=dev) for k in Ks]
    n = len(sets)
    L.check(lib.loft_roi_align_bwd_multi(
        L.arr(c_void_p, [g.data_ptr() for g in grad_feats]), L.arr(c_int, [s[2] for s in feat_shapes]),
        L.arr(c_int, [s[3] for s in feat_shapes]), L.arr(c_float, [1.0 / s for s in strides]), len(grad_feats),
        int(finest_scale), int(gos[0].shape[1]), L.dtype_code(gos[0]), n,
        L.arr(c_void_p, [r.data_ptr() for r in rois]), L.arr(c_int, Ks), L.arr(c_int, [int(s[2]) for s in sets]),
        L.arr(c_int, [int(s[3]) for s in sets]), L.arr(c_void_p, [g.data_ptr() for g in gos]), int(feat_shapes[0][0]),
        int(accumulate), L.arr(c_int, [int(bool(s[4])) for s in sets]), L.arr(c_void_p, [w.data_ptr() for w in wss]),
        L.dtype_code(grad_feats[0]), L.stream()), 'loft_roi_align_bwd_multi')
    return grad_feats


class _RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, P, strides, finest_scale, n_rot, *feats):
        ctx.save_for_backward(rois)
        ctx.meta = (P, tuple(strides), finest_scale, n_rot, [tuple(f.shape) for f in feats], feats[0].dtype)
        return roi_align_fwd(list(feats), rois, P, strides, finest_scale, n_rot)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        P, strides, fs, n_rot, shapes, dt = ctx.meta
        g = g.contiguous(memory_format=torch.channels_last)
        grads = roi_align_bwd(g, rois, shapes, P, strides, fs, n_rot,
                              out_dtype=dt if (dt == L.act16() and g.dtype == L.act16()) else torch.float32)
        return (None, None, None, None, None) + tuple(x.to(dt) for x in grads)


def roi_align(feats, rois, P, strides, finest_scale=56, n_rot=1):
    """Differentiable multi-level RoIAlign (SingleRoIExtractor.forward in one launch)."""
    return _RoIAlign.apply(rois, P, tuple(strides), finest_scale, n_rot, *feats)


def map_roi_levels(rois, num_levels=4, finest_scale=56):
    lib = L.load()
    L.dev_check(rois)
    rois = rois.float().contiguous()
    out = torch.empty(rois.shape[0], dtype=torch.int32, device=rois.device)
    L.check(lib.loft_map_roi_levels(L.ptr(rois), rois.shape[0], num_levels, finest_scale, L.ptr(out), L.stream()),
            'loft_map_roi_levels')
    return out


# ------------------------------------------------------------------ NMS / sort

NMS_PREDICATES = {'device': 0, 'cpu': 1}   # include/loft_hip.h LOFT_NMS_PRED_*


def nms_segmented(boxes_sorted, seg_offsets, iou_thr, seg_shift=None, max_segment=None, predicate='device', img_max=None, levels=1,
                  covered=False):
    """boxes_sorted [T,4] fp32 sorted (score desc, index asc) inside each segment;
    seg_offsets int64 [S+1] (device).  -> keep mask uint8 [T].
    predicate: 'device' (default; mmcv-1.0.5's CUDA kernel, inter > thr*union -- what the reference's GPU runs execute) or
    'cpu' (its host nms_cpu, inter/union >= thr); they differ exactly at IoU == thr."""
    lib = L.load()
    L.dev_check(boxes_sorted, seg_offsets, seg_shift)
    boxes_sorted = boxes_sorted.float().contiguous()
    T = boxes_sorted.shape[0]
    S = seg_offsets.numel() - 1
    # (covered: the caller's segments span all T boxes -- the scan writes every entry, no zero fill)
    keep = (torch.empty if covered and T > 0 and S > 0 else torch.zeros)(T, dtype=torch.uint8, device=boxes_sorted.device)
    if T == 0 or S <= 0:
        return keep
    if max_segment is None:
        max_segment = int((seg_offsets[1:] - seg_offsets[:-1]).max().item())
    ws = torch.empty(lib.loft_nms_workspace_bytes(T, max_segment, S), dtype=torch.uint8, device=boxes_sorted.device)
    if img_max is not None:     # segments = (image, level), shift = level * (img_max[image] + 1) computed on the device
        L.dev_check(img_max)
        L.check(lib.loft_nms_segmented_levels(L.ptr(boxes_sorted), L.ptr(seg_offsets), L.ptr(img_max), int(levels), S, c_int64(T),
                                              c_int64(max_segment), c_float(iou_thr), c_int(NMS_PREDICATES[predicate]), L.ptr(ws),
                                              L.ptr(keep), L.stream()), 'loft_nms_segmented_levels')
        return keep
    L.check(lib.loft_nms_segmented_pred(L.ptr(boxes_sorted), L.ptr(seg_offsets), L.ptr(seg_shift), S, c_int64(T),
                                        c_int64(max_segment), c_float(iou_thr), c_int(NMS_PREDICATES[predicate]), L.ptr(ws),
                                        L.ptr(keep), L.stream()), 'loft_nms_segmented_pred')
    return keep


def segmented_sort_desc(keys, seg_offsets, values=None):
    """Stable descending sort inside each segment -> (sorted_keys, sorted_values int32)."""
    lib = L.load()
    L.dev_check(keys, seg_offsets)
    keys = keys.float().contiguous()
    n = keys.numel()
    S = seg_offsets.numel() - 1
    if values is None:
        values = torch.arange(n, dtype=torch.int32, device=keys.device)
    values = values.to(torch.int32).contiguous()
    ko, vo = torch.empty_like(keys), torch.empty_like(values)
    if n == 0:
        return ko, vo
    nbytes = c_int64(0)
    L.check(lib.loft_segmented_sort_desc(L.ptr(keys), L.ptr(ko), L.ptr(values), L.ptr(vo), c_int64(n), S,
                                         L.ptr(seg_offsets), c_void_p(0), ctypes.byref(nbytes), L.stream()),
            'loft_segmented_sort_desc(query)')
    ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8, device=keys.device)
    L.check(lib.loft_segmented_sort_desc(L.ptr(keys), L.ptr(ko), L.ptr(values), L.ptr(vo), c_int64(n), S,
                                         L.ptr(seg_offsets), L.ptr(ws), ctypes.byref(nbytes), L.stream()),
            'loft_segmented_sort_desc')
    return ko, vo


TOPK_MAX = 4096     # nms.hip TOPK_MAX: candidates of one segment sorted in one workgroup's LDS


TOPK_MAX_SEGMENT = 32768   # one workgroup streams a whole segment four times: measured (tools/probes/topk_time.py) 33 us against the
                           # full sort's 63 us for 8 x 12 768 keys, but 330 us against 220 us for 8 x (196 608 + ...) anchors


_TOPK_TABLES = {}


def _topk_two_stage_tables(seg_lengths, k, device):
    """Static tables of the two-stage selection for segments longer than TOPK_MAX_SEGMENT: every segment is cut into sub-ranges
    ("runs") of at most `sub` keys (a multiple of 4: 16-byte loads), ~ TOPK_RUNS per longest segment.
    -> dict of device tables (run boundaries in the key array and in the compact candidate list, each run's segment), or None."""
    key = (tuple(int(n) for n in seg_lengths), int(k), str(device))
    if key in _TOPK_TABLES:
        return _TOPK_TABLES[key]
    longest = max(seg_lengths) if seg_lengths else 0
    sub = -(-longest // TOPK_RUNS)
    sub = (sub + 3) // 4 * 4
    tabs = None
    if 0 < sub <= TOPK_MAX_SEGMENT:
        sub_off, run_off, run_first, run_count, run_out = [0], [0], [], [], []
        pos = 0
        for n in seg_lengths:
            first, cnt = len(run_first), len(range(0, n, sub))
            for a in range(0, n, sub):
                ln = min(sub, n - a)
                sub_off.append(pos + a + ln)
                run_off.append(run_off[-1] + min(k, ln))
                run_first.append(first); run_count.append(cnt); run_out.append(pos)
            pos += n
        tabs = dict(sub_off=h2d(sub_off, torch.int64, device), run_off=h2d(run_off, torch.int64, device),
                    run_first=h2d(run_first, torch.int32, device), run_count=h2d(run_count, torch.int32, device),
                    run_out=h2d(run_out, torch.int64, device), n_cand=run_off[-1], n_run=len(run_first),
                    max_run=min(k, sub))
    _TOPK_TABLES[key] = tabs
    return tabs


TOPK_RUNS = 10          # sub-ranges per longest segment in the two-stage form (tools/probes/topk_time.py)


def segmented_topk_desc(keys, seg_offsets, k, values=None, max_segment=None, seg_lengths=None, key_mask=None):
    """The first k entries of each segment's stable descending order, in the layout of segmented_sort_desc (entry r of segment s at
    seg_offsets[s] + r; entries past min(k, segment length) are left unwritten).  In-house radix select + LDS bitonic sort
    (loft_segmented_topk_desc), one workgroup per segment, for k <= TOPK_MAX and segments of at most TOPK_MAX_SEGMENT keys
    (``max_segment``: the caller's host-side bound); longer segments with host-known lengths (``seg_lengths``) in TWO stages -- the
    same kernel over ~TOPK_RUNS sub-ranges per segment into a compact candidate list, then a rank merge of those sorted runs
    (loft_topk_merge_runs); otherwise the full library sort.  key_mask (uint8 / bool per key): masked keys count as -1."""
    lib = L.load()
    if key_mask is not None:
        key_mask = key_mask.view(torch.uint8) if key_mask.dtype == torch.bool else key_mask
        if key_mask.dtype != torch.uint8 or key_mask.numel() != keys.numel():
            raise L.LoftHipError('key_mask: one uint8 / bool per key')
        L.dev_check(key_mask)
    two_stage = max_segment is not None and max_segment > TOPK_MAX_SEGMENT
    if k > TOPK_MAX or max_segment is None or (two_stage and (seg_lengths is None or values is not None or key_mask is not None)):
        if key_mask is not None:
            keys = torch.where(key_mask.view(torch.bool).view_as(keys), keys, -1.0)
        return segmented_sort_desc(keys, seg_offsets, values)
    L.dev_check(keys, seg_offsets)
    keys = keys.float().contiguous()
    S = seg_offsets.numel() - 1
    ko = torch.empty_like(keys)
    vo = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    if keys.numel() == 0 or S <= 0:
        return ko, vo
    if two_stage:
        tabs = _topk_two_stage_tables(seg_lengths, k, keys.device)
        if tabs is None or len(seg_lengths) != S:
            return segmented_sort_desc(keys, seg_offsets, values)
        ck = torch.empty(tabs['n_cand'], dtype=torch.float32, device=keys.device)
        cv = torch.empty(tabs['n_cand'], dtype=torch.int32, device=keys.device)
        L.check(lib.loft_segmented_topk_desc(L.ptr(keys), L.ptr(ck), c_void_p(0), L.ptr(cv), tabs['n_run'], L.ptr(tabs['sub_off']), int(k),
                                             L.ptr(tabs['run_off']), c_void_p(0), L.stream()), 'loft_segmented_topk_desc(stage 1)')
        L.check(lib.loft_topk_merge_runs(L.ptr(ck), L.ptr(cv), tabs['n_run'], tabs['max_run'], L.ptr(tabs['run_off']),
                                         L.ptr(tabs['run_first']), L.ptr(tabs['run_count']), L.ptr(tabs['run_out']), int(k), L.ptr(ko),
                                         L.ptr(vo), L.stream()), 'loft_topk_merge_runs')
        return ko, vo
    if values is not None:
        values = values.to(torch.int32).contiguous()
    L.check(lib.loft_segmented_topk_desc(L.ptr(keys), L.ptr(ko), L.ptr(values), L.ptr(vo), S, L.ptr(seg_offsets), int(k), c_void_p(0),
                                         L.ptr(key_mask), L.stream()), 'loft_segmented_topk_desc')
    return ko, vo


def nms(boxes, scores, iou_thr, predicate='device'):
    """mmcv.ops.nms contract: -> (dets [M,5], keep [M] int64 in score-descending order)."""
    n = boxes.shape[0]
    off = torch.tensor([0, n], dtype=torch.int64, device=boxes.device)
    _, order = segmented_sort_desc(scores, off)
    order = order.long()
    keep_mask = nms_segmented(boxes[order], off, iou_thr, max_segment=n, predicate=predicate)
    keep = order[keep_mask.bool()]
    return torch.cat([boxes[keep], scores[keep, None]], dim=1), keep


# ------------------------------------------------------------------ dense contractions (MFMA)

_ZERO_PAGES = {}

# bench.py's live roofline probe: when set to a list, every MFMA launch appends
# (family, algorithmic_flops, start_event, end_event) recorded on the launch stream.
PROFILE = None
ALGO_SCALE = 1.0   # callers that zero-pad a contraction (narrow heads) scale the counted FLOPs back to the real ones


def _prof_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(ev, family, flops, tag=None):
    if ev is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    PROFILE.append((family, float(flops) * ALGO_SCALE, ev, end, tag))


def zero_page(device):
    key = str(device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(512, dtype=torch.uint8, device=device)
        if torch.device(device).type == 'cuda':
            # every stream's launches read this page from now on: it must BE zero before the first of them runs, whichever
            # stream that is (the fill above is only ordered on the current one).  Once per process.
            torch.cuda.current_stream().synchronize()
    return _ZERO_PAGES[key]


def pack_w_fwd(w, dtype=None):
    """[Cout,Cin,R,S] (reference / checkpoint layout) -> bf16 [R*S, Cout, Cin]."""
    co, ci, r, s = w.shape
    return w.permute(2, 3, 0, 1).reshape(r * s, co, ci).to(dtype or L.act16()).contiguous()


def pack_w_dgrad(w):
    """[Cout,Cin,R,S] -> bf16 [R*S, Cin, Cout] (transposed for the data-gradient pass)."""
    co, ci, r, s = w.shape
    return w.permute(2, 3, 1, 0).reshape(r * s, ci, co).to(L.act16()).contiguous()


def unpack_dw(dwp, shape):
    """fp32 [R*S, Cout, Cin] -> [Cout,Cin,R,S]."""
    co, ci, r, s = shape
    return dwp.view(r, s, co, ci).permute(2, 3, 0, 1).contiguous()


def _bf16(t):
    if t.dtype != L.act16():
        raise L.LoftHipError(f'expected {L.act16()} (the 16-bit type of the loaded library), got {t.dtype}')
    return t


# include/loft_hip.h LOFT_CONV_*: kernel selector of loft_conv_tap_bf16_v.  0 = the library's shape heuristics (the shipped path).
# tests / tools set CONV_VARIANT to a code, or to a callable (groups, B, OH, OW, Cin, Cout, T, ss, os) -> code, to pin a template.
CONV_AUTO, CONV_PIPE256, CONV_T256_FAST, CONV_T256, CONV_T128_SINGLE, CONV_T128_FAST, CONV_T128, CONV_T128x64, CONV_PATCH64, \
    CONV_STREAM256, CONV_STREAM128, CONV_STREAM64, CONV_STREAM64N, CONV_ROLES256, CONV_STREAM256N, CONV_RING32, CONV_W4, CONV_XFIRST, CONV_LEAN, CONV_LEANX = range(20)
CONV_FLAG_NO_PIXMAJOR, CONV_FLAG_NO_NFAST, CONV_FLAG_NO_STAGED_OUT, CONV_FLAG_TAP_MAJOR, CONV_FLAG_NO_ROI_BLOCKS, CONV_FLAG_KROT = \
    0x100, 0x200, 0x400, 0x800, 0x10000, 0x20000
CONV_VARIANT = CONV_AUTO
# include/loft_hip.h LOFT_F32_*: contraction of the fp32 parity mode.  SPLIT6 (default): fp32 operands as three bf16 each, six bf16
# MFMA terms per product (24 mantissa bits, fp32 accumulation: fp32-grade); SPLIT3: two bf16 / three terms (16 bits: faster, meets
# 1e-3 on losses / features / detections, not on every gradient); EXACT: the fp32 MFMA (bit-for-bit an fmaf chain).
F32_SPLIT6, F32_EXACT, F32_SPLIT3 = 0, 1, 2
# Round 5: the same contraction on OPERAND PLANES through the software-pipelined 16-bit kernels (loft_conv_tap_planes /
# loft_conv_wgrad_planes; include/loft_hip.h).  PLANES_F16 (default): every fp32 tensor as two binary16 planes under a power-of-two
# scale (22 significant bits), three products per element pair on v_mfma_f32_32x32x16_f16 -- the binary16 BUILD of the library,
# mapped next to the bfloat16 one; PLANES_BF16: three bfloat16 planes (24 bits), six products.  Shapes the stream kernels do not
# serve (Cout % 128, Cin % 64, accumulating launches) take SPLIT6's kernels.
F32_PLANES_F16, F32_PLANES_BF16, F32_PLANES_F16X4 = 3, 4, 5      # (F16X4: the binary16 planes with the lo x lo product as a fourth term)
_PLANE_MODES = {F32_PLANES_F16: (torch.float16, ((1, 0), (0, 1), (0, 0))),
                F32_PLANES_BF16: (torch.bfloat16, ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))),
                F32_PLANES_F16X4: (torch.float16, ((1, 1), (1, 0), (0, 1), (0, 0)))}
F32_CONTRACT = F32_PLANES_F16
PLANES_STATS = {'planes': 0, 'fallback': 0}      # launches of the fp32 mode by path (tests / bench read it)
PLANES_DB_FUSED = True     # the bias gradient of a plane weight-gradient launch in that launch (False: a loft_colsum_f32 pass; A/B, tests)


def _f32_kernel_code():
    """The LOFT_F32_* code handed to loft_conv_tap_f32_v / loft_conv_wgrad_f32_v (a plane mode's fallback is SPLIT6)."""
    return F32_SPLIT6 if F32_CONTRACT in _PLANE_MODES else int(F32_CONTRACT)


_SLOT_POOLS = {}      # raw stream -> [buffer, words used]: pre-zeroed (absmax, counter) pairs, one memset per 2048 tensors and stream


def _amax_slot(device):
    """Two zeroed words for loft_absmax_split_planes_f32, from a pool filled ON THE CURRENT STREAM (the zero fill is ordered
    before every kernel that uses a slot only on the stream that issued it)."""
    key = (str(device), L.stream().value)
    sp = _SLOT_POOLS.get(key)
    if sp is None or sp[1] + 2 > sp[0].numel():
        sp = _SLOT_POOLS[key] = [torch.zeros(4096, dtype=torch.float32, device=device), 0]
    v = sp[0][sp[1]:sp[1] + 2]
    sp[1] += 2
    return v


AMAX_FROM_PRODUCER = True     # a plane launch's epilogue leaves max |out| with its output tensor; the next split skips its absmax pass


def _known_amax(x):
    """The absmax slot a plane launch attached to its output, if the tensor is still what that launch wrote (same storage, no
    version bump; raw in-place kernels drop the note themselves: _drop_amax)."""
    note = getattr(x, '_loft_amax', None)
    if note is not None and AMAX_FROM_PRODUCER and note[1] == x.data_ptr() and note[2] == x._version and note[3] == x.numel():
        return note[0]
    return None


def _drop_amax(t):
    if t is not None and getattr(t, '_loft_amax', None) is not None:
        t._loft_amax = None


# The running trainer's batched weight planes (PrepackRegistry.request_f32 / run): (address, element count) of an fp32 operand packing
# -> (planes, absmax slot), made at the start of the step by two launches for all convs.  None outside a trainer step.
WEIGHT_PLANES = None


def split_planes(x, dtype16):
    """fp32 tensor (dense) -> (planes [NP, numel] of dtype16, absmax device scalar | None): loft_absmax_split_planes_f32, or the
    split alone when the tensor's producer already measured its absmax."""
    if WEIGHT_PLANES is not None and dtype16 == torch.float16:
        hit = WEIGHT_PLANES.get((x.data_ptr(), x.numel()))
        if hit is not None:
            return hit
    lib = L.load_for(dtype16)
    n = x.numel()
    planes = torch.empty((lib.loft_planes_per_tensor(), n), dtype=dtype16, device=x.device)
    if dtype16 != torch.float16:
        L.check(lib.loft_split_planes_f32(L.ptr(x), c_int64(n), L.ptr(planes), c_void_p(0), L.stream()), 'loft_split_planes_f32')
        return planes, None
    amax = _known_amax(x)
    if amax is not None:
        L.check(lib.loft_split_planes_f32(L.ptr(x), c_int64(n), L.ptr(planes), L.ptr(amax), L.stream()), 'loft_split_planes_f32')
        return planes, amax
    amax = _amax_slot(x.device)
    L.check(lib.loft_absmax_split_planes_f32(L.ptr(x), c_int64(n), L.ptr(planes), L.ptr(amax), L.stream()), 'loft_absmax_split_planes_f32')
    return planes, amax


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


# A backward node splits its output gradient for the data gradient AND the weight gradient: inside a planes_scoped function the
# same tensor (storage pointer + shape + version; kept alive with its planes until the scope ends) is split once.
_SCOPES = []


def planes_scoped(fn):
    """Decorator for autograd backward functions of the fp32 parity mode: one split memo for the duration of the call."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        _SCOPES.append({})
        try:
            return fn(*a, **k)
        finally:
            _SCOPES.pop()
    return wrapper


# Planes of a forward convolution's INPUT are needed again by that layer's weight gradient: conv2d_fwd leaves them here, keyed by
# (storage pointer, shape, version, plane type), until the tensor dies (weakref finalizer) or the weight gradient takes them.
_FWD_PLANES = {}
FWD_PLANES_CACHE = True
PLANES_XSTREAM = {'waits': 0}     # memo / cache hits served to ANOTHER stream than the one that split (tests read it)


class _PlanesEntry:
    """Planes of one tensor + where they were made.  A memo hit from ANOTHER stream (the backbone's weight gradients run on
    WGRAD_STREAM, the data gradient of the same node on the main stream; the forward's cached planes are made on main and
    consumed -- and dropped -- by the weight gradient) must not read planes that the producing stream has only enqueued, and the
    caching allocator must not hand their block back to the producing stream's pool while the consumer still reads it
    (ADVICE r5): the consumer waits for the producer's event and records itself on the planes and the absmax slot."""
    __slots__ = ('val', 'stream', 'event', 'keep')

    def __init__(self, val, keep=None):
        self.val = val
        self.keep = keep
        self.stream = L.stream().value
        self.event = torch.cuda.Event()
        self.event.record()

    def get(self):
        if L.stream().value != self.stream:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.event)
            for t in self.val:
                if t is not None:
                    t.record_stream(cur)
            PLANES_XSTREAM['waits'] += 1
        return self.val


def _fwd_planes_put(x, dtype16, val):
    import weakref
    key = (x.data_ptr(), tuple(x.shape), x._version, dtype16)
    _FWD_PLANES[key] = _PlanesEntry(val)
    weakref.finalize(x, _FWD_PLANES.pop, key, None)


def _split_memo(x, dtype16, memo, take_fwd=False):
    """split_planes with a memo (the caller's dict, else the innermost planes_scoped call's, else none); take_fwd: use (and drop)
    the planes the forward pass left for this tensor.  Entries are stream-aware (_PlanesEntry)."""
    if take_fwd:
        hit = _FWD_PLANES.pop((x.data_ptr(), tuple(x.shape), x._version, dtype16), None)
        if hit is not None:
            return hit.get()
    if memo is None:
        memo = _SCOPES[-1] if _SCOPES else None
    if memo is None:
        return split_planes(x, dtype16)
    key = (x.data_ptr(), tuple(x.shape), x._version, dtype16)
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = _PlanesEntry(split_planes(x, dtype16), keep=x)
    return hit.get()
WGRAD_AUTO, WGRAD_STREAM256, WGRAD_T256, WGRAD_T128, WGRAD_RING128 = range(5)       # LOFT_WGRAD_*: kernel selector of loft_conv_wgrad_bf16_v
WGRAD_VARIANT = WGRAD_AUTO      # a code, or a callable (groups, B, OH, OW, Cin, Cout, T, ss, gos) -> code


def _wgrad_variant(*shape):
    return WGRAD_VARIANT(*shape) if callable(WGRAD_VARIANT) else WGRAD_VARIANT


def conv_tap(src, wgt, out, B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, taps, ss=1, os=1, oo=(0, 0), bias=None,
             residual=None, relu=False, accumulate=False, groups=1, src_gs=0, wgt_gs=0, out_gs=0, bias_gs=0, mask=None,
             planes_memo=None, planes_cache=False, head=None):
    """Raw launch of loft_conv_tap_bf16 (or loft_conv_tap_f32 when every operand is fp32: the forward-only parity mode).
    taps: list of (dy, dx, weight_tap_index).
    head: dict(w=[c4,256] 16-bit, b=fp32 [c4], out=fp32 NHWC [B,c4,OHf,OWf]) -- a narrow 1x1 head on the output, computed in the
    launch's epilogue when the library serves it (loft_conv_tap_bf16_head); head['fused'] tells the caller whether it did."""
    lib = L.load()
    L.dev_check(src, wgt, out, bias, residual)
    if src.dtype == torch.float32:
        for t in (wgt, out, residual, mask):
            if t is not None and t.dtype != torch.float32:
                raise L.LoftHipError(f'fp32 parity mode needs every operand in fp32, got {t.dtype}')
        T = len(taps)
        if (F32_CONTRACT in _PLANE_MODES and not accumulate and Cin % 64 == 0 and Cout % 128 == 0 and _dense(src) and _dense(wgt)
                and src.numel() % 8 == 0 and wgt.numel() % 8 == 0 and B * OH * OW > 0):
            dt16, terms = _PLANE_MODES[F32_CONTRACT]
            if T * len(terms) <= 64:
                plib = L.load_for(dt16)
                (xp, ax), (wp, aw) = _split_memo(src, dt16, planes_memo), split_planes(wgt, dt16)
                # the launch writes EVERY element of a dense `out` (no output stride / offset, all groups): its epilogue can vouch
                # for the tensor's absmax
                whole = (dt16 == torch.float16 and AMAX_FROM_PRODUCER and os == 1 and OHf == OH and OWf == OW and _dense(out)
                         and out.numel() == groups * B * OH * OW * Cout)
                oslot = _amax_slot(src.device) if whole else None
                _drop_amax(out)
                if planes_cache and FWD_PLANES_CACHE:
                    _fwd_planes_put(src, dt16, (xp, ax))
                e = plib.loft_conv_tap_planes(L.ptr(xp), L.ptr(wp), L.ptr(bias), L.ptr(residual), L.ptr(mask), L.ptr(out),
                                              L.ptr(zero_page(src.device)), B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo[0], oo[1], ss, T,
                                              L.arr(c_int, [t[0] for t in taps]), L.arr(c_int, [t[1] for t in taps]),
                                              L.arr(c_int, [t[2] for t in taps]), int(relu), groups, c_int64(src_gs), c_int64(wgt_gs),
                                              c_int64(out_gs), c_int64(bias_gs), len(terms), L.arr(c_int, [t[0] for t in terms]),
                                              L.arr(c_int, [t[1] for t in terms]), c_int64(src.numel()), c_int64(wgt.numel()),
                                              L.ptr(ax), L.ptr(aw), L.ptr(oslot), L.stream())
                if e == 0:
                    PLANES_STATS['planes'] += 1
                    if oslot is not None:
                        out._loft_amax = (oslot, out.data_ptr(), out._version, out.numel())
                    return out
                if e != 1:              # (hipErrorInvalidValue: a shape the stream kernel does not serve -> the fp32 kernels below)
                    L.check(e, 'loft_conv_tap_planes')
        PLANES_STATS['fallback'] += 1
        _drop_amax(out)
        L.check(lib.loft_conv_tap_f32_v(L.ptr(src), L.ptr(wgt), L.ptr(bias), L.ptr(residual), L.ptr(mask), L.ptr(out),
                                        L.ptr(zero_page(src.device)), B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo[0],
                                        oo[1], ss, T, L.arr(c_int, [t[0] for t in taps]), L.arr(c_int, [t[1] for t in taps]),
                                        L.arr(c_int, [t[2] for t in taps]), int(relu), int(accumulate), groups,
                                        c_int64(src_gs), c_int64(wgt_gs), c_int64(out_gs), c_int64(bias_gs), _f32_kernel_code(),
                                        L.stream()), 'loft_conv_tap_f32_v')
        return out
    _bf16(src), _bf16(wgt)
    if residual is not None:
        _bf16(residual)
    T = len(taps)
    dy = L.arr(c_int, [t[0] for t in taps])
    dx = L.arr(c_int, [t[1] for t in taps])
    wt = L.arr(c_int, [t[2] for t in taps])
    out_f32 = out.dtype == torch.float32
    if not out_f32:
        _bf16(out)
    _ev = _prof_begin()
    variant = CONV_VARIANT(groups, B, OH, OW, Cin, Cout, T, ss, os) if callable(CONV_VARIANT) else CONV_VARIANT
    if head is not None:
        head['fused'] = False
        hw, hb, ho = head['w'], head['b'], head['out']
        if (HEAD_FUSION and not _DBG.no_head_fusion and variant == CONV_AUTO and groups == 1 and not out_f32 and not accumulate and Cout == 256 and
                hw.dtype == L.act16() and tuple(hw.shape[-2:]) == (int(ho.shape[1]), 256) and hw.is_contiguous() and
                ho.dtype == torch.float32 and tuple(ho.shape) == (B, int(ho.shape[1]), OHf, OWf) and
                ho.is_contiguous(memory_format=torch.channels_last)):
            e = lib.loft_conv_tap_bf16_head(L.ptr(src), L.ptr(wgt), L.ptr(bias), L.ptr(residual), L.ptr(mask), L.ptr(out),
                                            L.ptr(zero_page(src.device)), B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo[0], oo[1], ss, T,
                                            dy, dx, wt, int(relu), L.ptr(hw), L.ptr(hb.float().contiguous()), L.ptr(ho),
                                            int(ho.shape[1]), L.stream())
            if e == 0:
                head['fused'] = True
                _prof_end(_ev, 'conv_tap', 2.0 * B * OH * OW * Cout * (Cin * T + int(ho.shape[1])), (groups, B, OH, OW, Cin, Cout, T, ss, os))
                return out
            if e != 1:          # (hipErrorInvalidValue: a launch the head epilogue does not serve -> plain launch, head by the caller)
                L.check(e, 'loft_conv_tap_bf16_head')
    L.check(lib.loft_conv_tap_bf16_v(L.ptr(src), L.ptr(wgt), L.ptr(bias), L.ptr(residual), L.ptr(mask), L.ptr(out),
                                     L.ptr(zero_page(src.device)), B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo[0],
                                     oo[1], ss, T, dy, dx, wt, int(relu), int(out_f32), int(accumulate), groups,
                                     c_int64(src_gs), c_int64(wgt_gs), c_int64(out_gs), c_int64(bias_gs), int(variant),
                                     L.stream()), 'loft_conv_tap_bf16_v')
    _prof_end(_ev, 'conv_tap', 2.0 * groups * B * OH * OW * Cout * Cin * T, (groups, B, OH, OW, Cin, Cout, T, ss, os))
    return out


def conv_out_size(i, k, stride, pad):
    return (i + 2 * pad - k) // stride + 1


HEAD_FUSION = True     # tests / A/B: False = narrow heads always as launches of their own


def conv2d_fwd(x, wp, bias, R, S, stride=1, pad=0, relu=False, residual=None, out_dtype=None, groups=1, planes_cache=False,
               head=None):
    """x [G*B,Cin,IH,IW] channels_last bf16, wp [G][R*S,Cout,Cin] bf16 -> [G*B,Cout,OH,OW] channels_last.
    head = (packed head weight [.., c4, 256] 16-bit, bias fp32 [c4]): -> (out, head output fp32 NHWC [B,c4,OH,OW] | None): the
    narrow 1x1 head on `out`, from the same launch when the library serves it (None: the caller launches it)."""
    x = _nhwc(x)
    GB, Cin, IH, IW = x.shape
    B = GB // groups
    Cout = wp.shape[-2]
    OH, OW = conv_out_size(IH, R, stride, pad), conv_out_size(IW, S, stride, pad)
    out = empty_nhwc(GB, Cout, OH, OW, out_dtype or L.act16(), x.device)
    taps = [(r - pad, s - pad, r * S + s) for r in range(R) for s in range(S)]
    hd = None
    if head is not None and groups == 1 and x.dtype == L.act16():
        hw, hb = head
        c4 = int(hw.shape[-2])
        hd = dict(w=hw.reshape(c4, -1), b=hb, out=empty_nhwc(B, c4, OH, OW, torch.float32, x.device))
    conv_tap(x, wp, out, B, IH, IW, Cin, Cout, OH, OW, OH, OW, taps, ss=stride, bias=bias, residual=residual,
             relu=relu, groups=groups, src_gs=B * IH * IW * Cin, wgt_gs=R * S * Cout * Cin,
             out_gs=B * OH * OW * Cout, bias_gs=Cout, planes_cache=planes_cache, head=hd)
    if head is not None:
        return out, (hd['out'] if hd is not None and hd['fused'] else None)
    return out


def deconv2x2_fwd(x, wp, bias, out, relu=True, head=None):
    """ConvTranspose2d(k=2, s=2) + bias (+ ReLU) as ONE launch (loft_deconv2x2_bf16): x 16-bit NHWC [B,Cin,H,W], wp 16-bit [4,256,Cin]
    (tap p = 2 py + px), bias fp32 [256], out 16-bit NHWC [B,256,2H,2W]; head: the dict of conv_tap (narrow 1x1 head on the output).
    -> True when the library served it (else nothing was launched: the caller runs the four parity launches)."""
    lib = L.load()
    L.dev_check(x, wp, out, bias)
    B, Cin, H, W = x.shape
    Cout = int(wp.shape[-2])
    if (_DBG.no_deconv_fusion or CONV_VARIANT != CONV_AUTO or PROFILE is not None or x.dtype != L.act16() or out.dtype != L.act16()
            or tuple(wp.shape) != (4, Cout, Cin) or not wp.is_contiguous() or tuple(out.shape) != (B, Cout, 2 * H, 2 * W)):
        return False
    x, out = _nhwc(x), _nhwc(out)
    hw = hb = ho = None
    c4 = 0
    if head is not None:
        head['fused'] = False
        hw, hb, ho = head['w'], head['b'].float().contiguous(), head['out']
        c4 = int(ho.shape[1])
        if not (HEAD_FUSION and not _DBG.no_head_fusion and hw.dtype == L.act16() and tuple(hw.shape[-2:]) == (c4, 256)
                and hw.is_contiguous() and ho.dtype == torch.float32 and tuple(ho.shape) == (B, c4, 2 * H, 2 * W)
                and ho.is_contiguous(memory_format=torch.channels_last)):
            return False
    e = lib.loft_deconv2x2_bf16(L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(out), L.ptr(zero_page(x.device)), B, H, W, Cin, Cout, int(relu),
                                L.ptr(hw), L.ptr(hb), L.ptr(ho), c4, L.stream())
    if e == 1:
        return False
    L.check(e, 'loft_deconv2x2_bf16')
    if head is not None:
        head['fused'] = True
    return True


def bottleneck_tail(t1, wp2, b2, wp3, b3, shortcut, wpd=None, bd=None):
    """Tail of a 64-plane bottleneck in one launch (loft_bneck_tail_bf16): relu(W3 . relu(conv3x3(t1) + b2) + b3 + shortcut).
    t1 [B,64,H,W] channels_last 16-bit; wp2 [1][9,64,64], wp3 [1][1,256,64] forward packings; b2 [1][64], b3 [1][256] fp32;
    shortcut: the block input [B,256,H,W] (identity), or with wpd [1][1,256,64] / bd [1][256] the 64-channel block input x of a
    conv shortcut computed inside the launch (rounded to the 16-bit type before the add, as between launches)."""
    lib = L.load()
    t1 = _nhwc(t1)
    shortcut = _nhwc(shortcut)
    L.dev_check(t1, wp2, wp3, shortcut, wpd)
    B, C, H, W = t1.shape
    if C != 64 or wp2.shape[-3:] != (9, 64, 64) or wp3.shape[-3:] != (1, 256, 64) or shortcut.shape[1] != (64 if wpd is not None else 256):
        raise L.LoftHipError(f'bottleneck_tail: shapes {tuple(t1.shape)} {tuple(wp2.shape)} {tuple(wp3.shape)} {tuple(shortcut.shape)}')
    _bf16(t1), _bf16(wp2), _bf16(wp3), _bf16(shortcut)
    out = empty_nhwc(B, 256, H, W, L.act16(), t1.device)
    taps = [(r - 1, s - 1, r * 3 + s) for r in range(3) for s in range(3)]
    _ev = _prof_begin()
    L.check(lib.loft_bneck_tail_bf16(L.ptr(t1), L.ptr(wp2), L.ptr(b2), L.ptr(wp3), L.ptr(b3), L.ptr(shortcut), L.ptr(wpd), L.ptr(bd), L.ptr(out),
                                     L.ptr(zero_page(t1.device)), B, H, W, L.arr(c_int, [t[0] for t in taps]),
                                     L.arr(c_int, [t[1] for t in taps]), L.arr(c_int, [t[2] for t in taps]), L.stream()),
            'loft_bneck_tail_bf16')
    # (bench.py's live roofline: counted with the tap-conv family whose three / four launches it replaces)
    _prof_end(_ev, 'conv_tap', 2.0 * B * H * W * (9 * 64 * 64 + 64 * 256 * (2 if wpd is not None else 1)), (1, B, H, W, 64, 256, 9, 1, 1))
    return out


def bneck_pair_ok(a_in, C):
    """Shapes loft_bneck_pair_bf16 serves: a_in [B,P,H,W] channels_last 16-bit with P in {128, 256}, B*H*W % 128 == 0, C % 128 == 0."""
    B, P, H, W = a_in.shape
    return (a_in.is_cuda and a_in.dtype == L.act16() and P in (128, 256) and (B * H * W) % 128 == 0 and C % 128 == 0 and C >= 128
            and PROFILE is None)


def pack_k8(mats, outs=None, desc=None):
    """16-bit matrices [.., R, K] -> their K8 layouts [K/8, R, 8] (loft_pack_k8_multi: one launch for all of them).
    outs / desc: the buffers and the device descriptor table of a previous call on the SAME tensors (PrepackRegistry.run)."""
    lib = L.load()
    if outs is None:
        outs = [torch.empty(m.shape[-1] // 8, m.shape[-2], 8, dtype=m.dtype, device=m.device) for m in mats]
    if desc is None:
        for m in mats:
            L.dev_check(m)
            if m.shape[-1] % 8 or not m.is_contiguous() or m.numel() != m.shape[-1] * m.shape[-2] or m.dtype != L.act16():
                raise L.LoftHipError(f'pack_k8: {tuple(m.shape)} {m.dtype}')
        desc = h2d([[m.data_ptr(), o.data_ptr(), m.shape[-2], m.shape[-1]] for m, o in zip(mats, outs)], torch.int64, mats[0].device)
    L.check(lib.loft_pack_k8_multi(L.ptr(desc), len(mats), c_int64(max(m.numel() // 8 for m in mats)), L.stream()), 'loft_pack_k8_multi')
    return outs, desc


def bneck_pair(a_in, w1k8, bias1, res, w2k8, bias2, mask1=None, mask2=None, variant=0):
    """End of bottleneck k + start of bottleneck k+1 in one launch (loft_bneck_pair_bf16; see include/loft_hip.h).  Weights in the
    K8 layout (pack_k8 / PrepackRegistry.k8) of the usual packings:
    forward:  a_in = t2_k [B,P,H,W], w1k8 [P/8,C,8] / bias1 [..,C] = conv3_k (BN folded), res = the shortcut [B,C,H,W],
              w2k8 [C/8,P,8] / bias2 [..,P] = conv1_{k+1}  ->  (out_k [B,C,H,W], t1_{k+1} [B,P,H,W])
    backward: a_in = d t1_{k+1}, w1k8 = K8 of conv1_{k+1}'s data-gradient packing [C,P], res = the shortcut's gradient, mask1 = out_k,
              w2k8 = K8 of conv3_k's data-gradient packing [P,C], mask2 = t2_k, no biases  ->  (d out_k, d t2_k), both masked."""
    lib = L.load()
    a_in, res = _nhwc(a_in), _nhwc(res)
    B, P, H, W = a_in.shape
    C = res.shape[1]
    L.dev_check(a_in, w1k8, res, w2k8, mask1, mask2)
    _bf16(a_in), _bf16(w1k8), _bf16(res), _bf16(w2k8)
    if tuple(w1k8.shape) != (P // 8, C, 8) or tuple(w2k8.shape) != (C // 8, P, 8) or tuple(res.shape) != (B, C, H, W) \
            or not (w1k8.is_contiguous() and w2k8.is_contiguous()):
        raise L.LoftHipError(f'bneck_pair: shapes {tuple(a_in.shape)} {tuple(w1k8.shape)} {tuple(res.shape)} {tuple(w2k8.shape)}')
    if mask1 is not None:
        mask1, mask2 = _nhwc(mask1), _nhwc(mask2)
        _bf16(mask1), _bf16(mask2)
        if tuple(mask1.shape) != (B, C, H, W) or tuple(mask2.shape) != (B, P, H, W):
            raise L.LoftHipError(f'bneck_pair: mask shapes {tuple(mask1.shape)} {tuple(mask2.shape)}')
    mid = empty_nhwc(B, C, H, W, L.act16(), a_in.device)
    out2 = empty_nhwc(B, P, H, W, L.act16(), a_in.device)
    L.check(lib.loft_bneck_pair_bf16_v(L.ptr(a_in), L.ptr(w1k8), L.ptr(bias1), L.ptr(res), L.ptr(mask1), L.ptr(mid), L.ptr(w2k8),
                                       L.ptr(bias2), L.ptr(mask2), L.ptr(out2), c_int64(B * H * W), P, C, int(variant), L.stream()),
            'loft_bneck_pair_bf16')
    return mid, out2


def conv2d_dgrad(g, wpt, in_hw, R, S, stride=1, pad=0, residual=None, out_dtype=None, groups=1,
                 out=None, accumulate=False, mask=None):
    """g [G*B,Cout,OH,OW] channels_last bf16, wpt [G][R*S,Cin,Cout] -> grad of the conv input [G*B,Cin,IH,IW]."""
    g = _nhwc(g)
    GB, Cout, OH, OW = g.shape
    B = GB // groups
    Cin = wpt.shape[-2]
    IH, IW = in_hw
    if out is None:
        out = empty_nhwc(GB, Cin, IH, IW, out_dtype or L.act16(), g.device)
    gs = dict(groups=groups, src_gs=B * OH * OW * Cout, wgt_gs=R * S * Cin * Cout, out_gs=B * IH * IW * Cin)
    if stride == 1:
        taps = [(pad - r, pad - s, r * S + s) for r in range(R) for s in range(S)]
        conv_tap(g, wpt, out, B, OH, OW, Cout, Cin, IH, IW, IH, IW, taps, residual=residual, accumulate=accumulate,
                 mask=mask, **gs)
        return out
    # strided: one launch per output-parity class, each with the taps that reach it
    covered = 0
    launches = []
    for py in range(stride):
        for px in range(stride):
            taps = [((py + pad - r) // stride, (px + pad - s) // stride, r * S + s)
                    for r in range(R) for s in range(S)
                    if (py + pad - r) % stride == 0 and (px + pad - s) % stride == 0]
            nh, nw = (IH - py + stride - 1) // stride, (IW - px + stride - 1) // stride
            if taps and nh > 0 and nw > 0:
                launches.append((py, px, taps, nh, nw))
                covered += 1
    if covered < stride * stride and not accumulate:
        if residual is not None:
            # (out IS the residual buffer: the positions no parity class reaches already hold it, and a covered position's residual
            #  is read by the lane that then overwrites it -- no full-map copy in front of a launch that touches a quarter of it)
            if out.data_ptr() != residual.data_ptr():
                out.copy_(residual)
        else:
            out.zero_()
        residual_for_launch = residual
    else:
        residual_for_launch = residual
    for py, px, taps, nh, nw in launches:
        conv_tap(g, wpt, out, B, OH, OW, Cout, Cin, nh, nw, IH, IW, taps, ss=1, os=stride, oo=(py, px),
                 residual=residual_for_launch, accumulate=accumulate, mask=mask, **gs)
    return out


import os as _os


# Split-K combination of the weight gradients whose consumer is the batched unpack: False = fp32 atomics into a zeroed buffer
# (shipped), True = per-split slots written with plain stores and summed by the unpack while it reads.  Measured (MI355X, bench
# step, same box): the slots take 1.2 ms of KERNEL time off a step in the serialised profile (conv_wgrad_kernel<128,4> 72 -> 57 us,
# stream kernels -35..-45 us per launch, unpack 0.39 -> 0.9 ms) but the step itself is unchanged at 30.2 ms -- with the branch /
# weight-gradient streams overlapped the atomic tails (L2-bound, no HBM, no MFMA) were already hidden behind other kernels, while
# 3 GB of slot stores + reads per step compete for HBM with them.  Kept selectable and tested; not the default.
# Round 5, re-measured on the round-4 tree (tools/ab_env.sh LOFT_BENCH_SLOTS=dense, same box, two rounds): slots for the DENSE-map
# launches only (B <= 64: backbone / FPN, ~50 launches of 288 workgroups whose 16k atomics per workgroup serialise at the end of a
# single-round launch) 34.64 / 34.66 ms against 35.00 / 34.87 -- their slot traffic is 0.9 GB per step, not 3 -- so that is the
# default; the RoI-map launches (long K, atomics a few percent) keep the atomics.
WGRAD_SLOTS = lambda G, B, OH, OW, Cin, Cout, T, ss, gos: B <= 64      # noqa: E731


def conv_wgrad(g, x, B, GH, GW, Cout, XH, XW, Cin, OH, OW, taps, n_wtaps, gos=1, ss=1, groups=1, g_gs=0, x_gs=0,
               splits=0, dw=None, db=None, db_tap=-1, slots_ok=False, planes_memo=None):
    """Raw launch of loft_conv_wgrad_bf16.  taps: list of (goy, gox, dy, dx, weight_tap_index).
    db: optional zeroed fp32 [groups, Cout] -> bias gradient accumulated in the same pass.
    slots_ok: the caller sums split-K slots itself (UnpackQueue): the result may then be fp32 [groups, S, n_wtaps, Cout, Cin],
    every slot written with plain stores (loft_conv_wgrad_bf16_slots), instead of [groups, n_wtaps, Cout, Cin]."""
    lib = L.load()
    L.dev_check(g, x)
    A = lambda i: L.arr(c_int, [t[i] for t in taps])
    if g.dtype == torch.float32 and x.dtype == torch.float32:
        # fp32 parity mode (parity_f32.hip; F32_CONTRACT); the bias gradient is the plain column sum of g
        if dw is None:
            dw = torch.zeros(groups, n_wtaps, Cout, Cin, dtype=torch.float32, device=g.device)
        done = False
        if (F32_CONTRACT in _PLANE_MODES and Cout % 128 == 0 and Cin % 128 == 0 and _dense(g) and _dense(x) and g.numel() % 8 == 0
                and x.numel() % 8 == 0 and B * OH * OW > 0 and groups * len(_PLANE_MODES[F32_CONTRACT][1]) <= 32):
            dt16, terms = _PLANE_MODES[F32_CONTRACT]
            plib = L.load_for(dt16)
            (gp, ag), (xp, ax) = _split_memo(g, dt16, planes_memo), _split_memo(x, dt16, planes_memo, take_fwd=True)
            # the bias gradient rides in the same launch (the all-ones MFMA of the 16-bit kernels on every G plane; round 6) when
            # db is the plain [groups, Cout] accumulator and the caller named a tap whose G rows are complete
            db_fused = (PLANES_DB_FUSED and db is not None and db_tap != -1 and tuple(db.shape) == (groups, Cout) and db.is_contiguous()
                        and db.dtype == torch.float32)
            e = plib.loft_conv_wgrad_planes(L.ptr(gp), L.ptr(xp), L.ptr(dw), L.ptr(zero_page(g.device)), B, GH, GW, Cout, XH, XW, Cin,
                                            OH, OW, gos, ss, len(taps), A(0), A(1), A(2), A(3), A(4), groups, c_int64(g_gs),
                                            c_int64(x_gs), c_int64(n_wtaps * Cout * Cin), len(terms),
                                            L.arr(c_int, [t[1] for t in terms]), L.arr(c_int, [t[0] for t in terms]),
                                            c_int64(g.numel()), c_int64(x.numel()), L.ptr(ag), L.ptr(ax),
                                            L.ptr(db) if db_fused else c_void_p(0), int(db_tap) if db_fused else -1, L.stream())
            if e == 0:
                PLANES_STATS['planes'] += 1
                done = True
                if db_fused:
                    PLANES_STATS['db_fused'] = PLANES_STATS.get('db_fused', 0) + 1
                    db = None                       # (done: no column-sum pass below)
            elif e != 1:
                L.check(e, 'loft_conv_wgrad_planes')
        if not done:
            PLANES_STATS['fallback'] += 1
            L.check(lib.loft_conv_wgrad_f32_v(L.ptr(g), L.ptr(x), L.ptr(dw), B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, len(taps),
                                          A(0), A(1), A(2), A(3), A(4), groups, c_int64(g_gs), c_int64(x_gs),
                                              c_int64(n_wtaps * Cout * Cin), _f32_kernel_code(), L.stream()), 'loft_conv_wgrad_f32_v')
        if db is not None:
            if _dense(g) and g.dim() == 4 and Cout % 4 == 0 and g.is_contiguous(memory_format=torch.channels_last):
                tmp = db if (db.shape[1] == Cout and db.is_contiguous()) else torch.zeros(groups, Cout, dtype=torch.float32, device=g.device)
                L.check(lib.loft_colsum_f32(L.ptr(g), c_int64(B * GH * GW), Cout, groups, L.ptr(tmp), L.stream()), 'loft_colsum_f32')
                if tmp is not db:
                    db += tmp[:, :db.shape[1]]
            else:
                db += g.view(groups, -1, *g.shape[1:]).sum(dim=(1, 3, 4))[:, :db.shape[1]]
        return dw
    _bf16(g), _bf16(x)
    _ev = _prof_begin()
    if (Cout <= 64 and Cin <= 64 and len(taps) <= 9 and gos == 1 and ss == 1 and (GH, GW) == (OH, OW) == (XH, XW)
            and B * OH * OW >= 65536 and all(t[0] == 0 and t[1] == 0 and abs(t[2]) <= 1 and abs(t[3]) <= 1 for t in taps)
            and (db is None or db_tap != -1) and not _DBG.wgrad_no_patch):
        # narrow stride-1 convs at high resolution: all taps from one staged pixel patch (loft_conv_wgrad_patch_bf16)
        if dw is None:
            dw = pooled_zeros((groups, n_wtaps, Cout, Cin), g.device)
        ws = torch.empty(lib.loft_conv_wgrad_patch_workspace_bytes(B, OH, OW, Cout, Cin, len(taps), groups), dtype=torch.uint8,
                         device=g.device)
        L.check(lib.loft_conv_wgrad_patch_bf16(L.ptr(g), L.ptr(x), L.ptr(dw), L.ptr(zero_page(g.device)), B, OH, OW, Cout, Cin,
                                               len(taps), A(2), A(3), A(4), groups, c_int64(g_gs), c_int64(x_gs),
                                               c_int64(n_wtaps * Cout * Cin), L.ptr(db), L.ptr(ws), L.stream()),
                'loft_conv_wgrad_patch_bf16')
        _prof_end(_ev, 'conv_wgrad', 2.0 * groups * B * OH * OW * Cout * Cin * len(taps), (groups, B, OH, OW, Cin, Cout, len(taps), ss, gos))
        return dw
    use_slots = WGRAD_SLOTS(groups, B, OH, OW, Cin, Cout, len(taps), ss, gos) if callable(WGRAD_SLOTS) else WGRAD_SLOTS
    if slots_ok and use_slots and dw is None and n_wtaps == len(taps):
        S = lib.loft_conv_wgrad_slots(B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, len(taps), A(0), A(1), A(2), A(3), A(4),
                                      groups, splits, int(_wgrad_variant(groups, B, OH, OW, Cin, Cout, len(taps), ss, gos)))
        if S < 0:
            L.check(-S, 'loft_conv_wgrad_slots')
        if S >= 1:
            dw = pooled_scratch((groups, S, n_wtaps, Cout, Cin), g.device)
            L.check(lib.loft_conv_wgrad_bf16_slots(L.ptr(g), L.ptr(x), L.ptr(dw), L.ptr(zero_page(g.device)), B, GH, GW, Cout, XH,
                                                   XW, Cin, OH, OW, gos, ss, len(taps), A(0), A(1), A(2), A(3), A(4), groups,
                                                   c_int64(g_gs), c_int64(x_gs), S, splits, L.ptr(db), int(db_tap),
                                                   int(_wgrad_variant(groups, B, OH, OW, Cin, Cout, len(taps), ss, gos)), L.stream()), 'loft_conv_wgrad_bf16_slots')
            _prof_end(_ev, 'conv_wgrad', 2.0 * groups * B * OH * OW * Cout * Cin * len(taps), (groups, B, OH, OW, Cin, Cout, len(taps), ss, gos))
            return dw
    if dw is None:
        dw = pooled_zeros((groups, n_wtaps, Cout, Cin), g.device)
    L.check(lib.loft_conv_wgrad_bf16_v(L.ptr(g), L.ptr(x), L.ptr(dw), L.ptr(zero_page(g.device)), B, GH, GW, Cout, XH,
                                       XW, Cin, OH, OW, gos, ss, len(taps), A(0), A(1), A(2), A(3), A(4), groups,
                                       c_int64(g_gs), c_int64(x_gs), c_int64(n_wtaps * Cout * Cin), splits, L.ptr(db),
                                       int(db_tap), int(_wgrad_variant(groups, B, OH, OW, Cin, Cout, len(taps), ss, gos)), L.stream()),
            'loft_conv_wgrad_bf16_v')
    _prof_end(_ev, 'conv_wgrad', 2.0 * groups * B * OH * OW * Cout * Cin * len(taps), (groups, B, OH, OW, Cin, Cout, len(taps), ss, gos))
    return dw


def conv2d_wgrad(g, x, R, S, stride=1, pad=0, groups=1, splits=0, with_bias=False, slots_ok=False):
    """-> fp32 [G, R*S, Cout, Cin] (packed layout; see unpack_dw); with_bias: also -> fp32 [G, Cout] bias gradient.
    slots_ok: see conv_wgrad (-> possibly [G, S, R*S, Cout, Cin])."""
    g, x = _nhwc(g), _nhwc(x)
    GB, Cout, OH, OW = g.shape
    _, Cin, IH, IW = x.shape
    B = GB // groups
    taps = [(0, 0, r - pad, s - pad, r * S + s) for r in range(R) for s in range(S)]
    db, db_tap = None, -1
    if with_bias:
        centre = [i for i, t in enumerate(taps) if t[2] == 0 and t[3] == 0]
        if not centre:
            raise L.LoftHipError('fused bias gradient needs a tap with zero offset')
        db, db_tap = pooled_zeros((groups, Cout), g.device), centre[0]
    dw = conv_wgrad(g, x, B, OH, OW, Cout, IH, IW, Cin, OH, OW, taps, R * S, gos=1, ss=stride, groups=groups,
                    g_gs=B * OH * OW * Cout, x_gs=B * IH * IW * Cin, splits=splits, db=db, db_tap=db_tap, slots_ok=slots_ok)
    return (dw, db) if with_bias else dw


# ------------------------------------------------------------------ HBM-bound glue

def relu_bwd(g, y):
    lib = L.load()
    L.dev_check(g, y)
    out = torch.empty_like(g)
    if g.dtype == torch.float32 and y.dtype == torch.float32:          # fp32 parity mode
        # (the masked gradient is split into operand planes next: the kernel leaves its absmax with it -- kernels._known_amax)
        slot = _amax_slot(g.device) if (AMAX_FROM_PRODUCER and F32_CONTRACT in (F32_PLANES_F16, F32_PLANES_F16X4)) else None
        L.check(lib.loft_relu_bwd_f32(L.ptr(g), L.ptr(y), L.ptr(out), c_int64(g.numel()), L.ptr(slot), L.stream()), 'loft_relu_bwd_f32')
        if slot is not None:
            out._loft_amax = (slot, out.data_ptr(), out._version, out.numel())
        return out
    L.check(lib.loft_relu_bwd_bf16(L.ptr(_bf16(g)), L.ptr(_bf16(y)), L.ptr(out), c_int64(g.numel()), L.stream()),
            'loft_relu_bwd_bf16')
    return out


def colsum(x2d_like, C):
    """x: any bf16 tensor whose memory is [M][C] -> fp32 [C]."""
    lib = L.load()
    L.dev_check(x2d_like)
    out = torch.zeros(C, dtype=torch.float32, device=x2d_like.device)
    L.check(lib.loft_colsum_bf16(L.ptr(_bf16(x2d_like)), c_int64(x2d_like.numel() // C), C, L.ptr(out), L.stream()),
            'loft_colsum_bf16')
    return out


def upsample2x_add_(fine, coarse):
    lib = L.load()
    fine, coarse = _nhwc(fine), _nhwc(coarse)
    B, C, H, W = fine.shape
    if fine.dtype == torch.float32 and coarse.dtype == torch.float32:
        _drop_amax(fine)            # (written in place by a raw kernel: a producer's absmax note no longer holds)
        L.check(lib.loft_upsample2x_add_f32(L.ptr(fine), L.ptr(coarse), B, H, W, C, L.stream()), 'loft_upsample2x_add_f32')
        return fine
    L.check(lib.loft_upsample2x_add_bf16(L.ptr(_bf16(fine)), L.ptr(_bf16(coarse)), B, H, W, C, L.stream()),
            'loft_upsample2x_add_bf16')
    return fine


def downsum2x_add_(coarse, fine):
    lib = L.load()
    fine, coarse = _nhwc(fine), _nhwc(coarse)
    B, C, Hc, Wc = coarse.shape
    if fine.dtype == torch.float32 and coarse.dtype == torch.float32:  # fp32 parity mode
        _drop_amax(coarse)
        L.check(lib.loft_downsum2x_add_f32(L.ptr(coarse), L.ptr(fine), B, Hc, Wc, fine.shape[2], fine.shape[3], C, L.stream()),
                'loft_downsum2x_add_f32')
        return coarse
    L.check(lib.loft_downsum2x_add_bf16(L.ptr(_bf16(coarse)), L.ptr(_bf16(fine)), B, Hc, Wc, C, L.stream()),
            'loft_downsum2x_add_bf16')
    return coarse


def downsum2x_sum(coarse, fine):
    """coarse + 2x2 block sums of fine as a NEW tensor (loft_downsum2x_sum_bf16; same additions in the same order as
    downsum2x_add_): 16-bit activations only."""
    lib = L.load()
    fine, coarse = _nhwc(fine), _nhwc(coarse)
    B, C, Hc, Wc = coarse.shape
    out = empty_nhwc(B, C, Hc, Wc, coarse.dtype, coarse.device)
    L.check(lib.loft_downsum2x_sum_bf16(L.ptr(out), L.ptr(_bf16(coarse)), L.ptr(_bf16(fine)), B, Hc, Wc, C, L.stream()),
            'loft_downsum2x_sum_bf16')
    return out


def subsample2(x):
    lib = L.load()
    x = _nhwc(x)
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = empty_nhwc(B, C, Ho, Wo, x.dtype, x.device)
    if x.dtype == torch.float32:
        L.check(lib.loft_subsample2_f32(L.ptr(x), L.ptr(out), B, Ho, Wo, H, W, C, L.stream()), 'loft_subsample2_f32')
        return out
    L.check(lib.loft_subsample2_bf16(L.ptr(_bf16(x)), L.ptr(out), B, Ho, Wo, H, W, C, 0, L.stream()),
            'loft_subsample2_bf16')
    return out


def subsample2_adjoint_add_(big, small):
    lib = L.load()
    big, small = _nhwc(big), _nhwc(small)
    B, C, H, W = big.shape
    if big.dtype == torch.float32 and small.dtype == torch.float32:    # fp32 parity mode
        _drop_amax(big)
        L.check(lib.loft_subsample2_add_f32(L.ptr(big), L.ptr(small), B, small.shape[2], small.shape[3], H, W, C, L.stream()),
                'loft_subsample2_add_f32')
        return big
    L.check(lib.loft_subsample2_bf16(L.ptr(_bf16(small)), L.ptr(_bf16(big)), B, small.shape[2], small.shape[3], H, W, C,
                                     1, L.stream()), 'loft_subsample2_bf16(adjoint)')
    return big


def maxpool3x3s2(x):
    lib = L.load()
    x = _nhwc(x)
    B, C, H, W = x.shape
    out = empty_nhwc(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.dtype, x.device)
    if x.dtype == torch.float32:
        L.check(lib.loft_maxpool3x3s2_f32(L.ptr(x), L.ptr(out), B, H, W, C, L.stream()), 'loft_maxpool3x3s2_f32')
        return out
    L.check(lib.loft_maxpool3x3s2_bf16(L.ptr(_bf16(x)), L.ptr(out), B, H, W, C, L.stream()), 'loft_maxpool3x3s2_bf16')
    return out


def stem7x7_bn_relu(img, w, scale, shift, out_dtype=None):
    """img fp32 NCHW [B,3,H,W] -> bf16 (or fp32: parity mode) channels_last [B,64,H/2,W/2]."""
    lib = L.load()
    L.dev_check(img, w, scale, shift)
    img = img.float().contiguous()
    w, scale, shift = w.float().contiguous(), scale.float().contiguous(), shift.float().contiguous()
    B, _, H, W = img.shape
    out_dtype = out_dtype or L.act16()
    out = empty_nhwc(B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1, out_dtype, img.device)
    L.check(lib.loft_stem7x7_bn_relu(L.ptr(img), L.ptr(w), L.ptr(scale),
                                     L.ptr(shift), L.ptr(out), B, H, W, int(out_dtype == torch.float32), L.stream()),
            'loft_stem7x7_bn_relu')
    return out


def stem7x7_pack(w, scale, shift):
    """-> (wp 16-bit [64,192]: w * BN scale, k = c*49 + r*7 + s zero-padded to 192; bias fp32 [64] = BN shift)."""
    wp = torch.zeros(64, 192, dtype=L.act16(), device=w.device)
    wp[:, :147] = (w.float() * scale.float()[:, None, None, None]).reshape(64, 147).to(L.act16())
    return wp, shift.float().contiguous()


def stem7x7_mfma(img, w, scale, shift, packed=None):
    """MFMA stem: img fp32 NCHW [B,3,H,W], w fp32 [64,3,7,7], folded BN scale/shift -> bf16 channels_last [B,64,H/2,W/2].
    packed = stem7x7_pack(...) of a frozen stem (cached by the caller): no packing launches."""
    lib = L.load()
    L.dev_check(img, w)
    img = img.float().contiguous()
    wp, bias = packed if packed is not None else stem7x7_pack(w, scale, shift)
    B, _, H, W = img.shape
    out = empty_nhwc(B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1, L.act16(), img.device)
    L.check(lib.loft_stem7x7_mfma(L.ptr(img), L.ptr(wp), L.ptr(bias), L.ptr(out), B, H, W, L.stream()),
            'loft_stem7x7_mfma')
    return out


def cast_bf16(x_f32):
    lib = L.load()
    L.dev_check(x_f32)
    out = torch.empty_like(x_f32, dtype=L.act16())
    L.check(lib.loft_cast_f32_to_bf16(L.ptr(x_f32), L.ptr(out), c_int64(x_f32.numel()), L.stream()),
            'loft_cast_f32_to_bf16')
    return out


def add_bf16(a, b):
    lib = L.load()
    L.dev_check(a, b)
    out = torch.empty_like(a)
    L.check(lib.loft_add_bf16(L.ptr(_bf16(a)), L.ptr(_bf16(b)), L.ptr(out), c_int64(a.numel()), L.stream()),
            'loft_add_bf16')
    return out


def sumsq_(g_flat, out):
    lib = L.load()
    L.check(lib.loft_sumsq_f32(L.ptr(g_flat), c_int64(g_flat.numel()), L.ptr(out), L.stream()), 'loft_sumsq_f32')
    return out


def sgd_momentum_(p, g, m, gnorm_sq, max_norm, lr, momentum, weight_decay, grad_scale=1.0):
    lib = L.load()
    L.dev_check(p, g, m, gnorm_sq)
    L.check(lib.loft_sgd_momentum_f32(L.ptr(p), L.ptr(g), L.ptr(m), c_int64(p.numel()), L.ptr(gnorm_sq),
                                      c_float(max_norm), c_float(lr), c_float(momentum), c_float(weight_decay),
                                      c_float(grad_scale), L.stream()), 'loft_sgd_momentum_f32')


# ------------------------------------------------------------------ boxes / targets

def iou_assign(boxes, nbox, gts, ngt, pos_thr, neg_thr, min_pos, low_quality=True):
    """boxes [B,N,4], nbox int32 [B], gts [B,K,4], ngt int32 [B] -> (gt_inds int64 [B,N], max_ov [B,N])."""
    lib = L.load()
    L.dev_check(boxes, nbox, gts, ngt)
    boxes, gts = boxes.float().contiguous(), gts.float().contiguous()
    B, N = boxes.shape[:2]
    K = gts.shape[1]
    dev = boxes.device
    max_ov = torch.empty(B, N, dtype=torch.float32, device=dev)
    argmax = torch.empty(B, N, dtype=torch.int32, device=dev)
    gtmax = torch.empty(B, max(K, 1), dtype=torch.int32, device=dev)
    gt_inds = torch.empty(B, N, dtype=torch.int64, device=dev)
    nbox_i, ngt_i = nbox.int().contiguous(), ngt.int().contiguous()  # keep alive across the launch
    L.check(lib.loft_iou_assign(L.ptr(boxes), L.ptr(nbox_i), N, L.ptr(gts), L.ptr(ngt_i), K, B,
                                c_float(pos_thr), c_float(neg_thr), c_float(min_pos), int(low_quality), L.ptr(max_ov),
                                L.ptr(argmax), L.ptr(gtmax), L.ptr(gt_inds), L.stream()), 'loft_iou_assign')
    return gt_inds, max_ov


def delta2bbox(rois, deltas, means, stds, max_shape=None, wh_ratio_clip=16 / 1000):
    lib = L.load()
    L.dev_check(rois, deltas)
    rois, deltas = rois.float().contiguous(), deltas.float().contiguous()
    out = torch.empty_like(rois)
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    L.check(lib.loft_delta2bbox(L.ptr(rois), L.ptr(deltas), c_int64(rois.shape[0]), L.arr(c_float, list(means)),
                                L.arr(c_float, list(stds)), c_float(wh_ratio_clip), c_float(mh), c_float(mw), L.ptr(out),
                                L.stream()), 'loft_delta2bbox')
    return out


def bbox2delta(proposals, gt, means, stds):
    lib = L.load()
    L.dev_check(proposals, gt)
    proposals, gt = proposals.float().contiguous(), gt.float().contiguous()
    out = torch.empty_like(proposals)
    L.check(lib.loft_bbox2delta(L.ptr(proposals), L.ptr(gt), c_int64(proposals.shape[0]), L.arr(c_float, list(means)),
                                L.arr(c_float, list(stds)), L.ptr(out), L.stream()), 'loft_bbox2delta')
    return out


def rpn_scores(head, A, img_stride, lvl_off, keys):
    lib = L.load()
    head = _nhwc(head)
    B, Cp, H, W = head.shape
    L.check(lib.loft_rpn_scores(L.ptr(head), B, H, W, Cp, A, c_int64(img_stride), c_int64(lvl_off), L.ptr(keys),
                                L.stream()), 'loft_rpn_scores')


def rpn_decode(head, sorted_idx, A, img_stride, lvl_off, topk, base_anchors, stride, means, stds, max_shape, cand_stride,
               cand_off, out_boxes, wh_ratio_clip=16 / 1000):
    lib = L.load()
    head = _nhwc(head)
    B, Cp, H, W = head.shape
    L.check(lib.loft_rpn_decode(L.ptr(head), L.ptr(sorted_idx), B, H, W, Cp, A, c_int64(img_stride), c_int64(lvl_off),
                                int(topk), L.ptr(base_anchors), int(stride), L.arr(c_float, list(means)),
                                L.arr(c_float, list(stds)), c_float(wh_ratio_clip), c_float(max_shape[0]),
                                c_float(max_shape[1]), c_int64(cand_stride), c_int64(cand_off), L.ptr(out_boxes),
                                L.stream()), 'loft_rpn_decode')


def rpn_scores_levels(heads, A, img_stride, lvl_offs, keys, img_max=None):
    """rpn_scores for every level in one launch; img_max (fp32 [B], optional) is reset to -inf for rpn_decode_levels."""
    lib = L.load()
    heads = [_nhwc(h) for h in heads]
    B, Cp = heads[0].shape[:2]
    n = len(heads)
    L.check(lib.loft_rpn_scores_levels(L.arr(c_void_p, [h.data_ptr() for h in heads]), L.arr(c_int, [int(h.shape[2]) for h in heads]),
                                       L.arr(c_int, [int(h.shape[3]) for h in heads]), L.arr(c_int64, [int(o) for o in lvl_offs[:n]]),
                                       n, B, Cp, A, c_int64(img_stride), L.ptr(keys), L.ptr(img_max), L.stream()),
            'loft_rpn_scores_levels')
    return heads


def rpn_decode_levels(heads, sorted_idx, sorted_keys, A, img_stride, lvl_offs, topk, base_anchors, strides, means, stds, max_shape,
                      cand_stride, cand_offs, out_boxes, out_scores=None, img_max=None, wh_ratio_clip=16 / 1000):
    """rpn_decode for every level in one launch (+ candidate scores in the candidate layout, + the per-image coordinate maximum)."""
    lib = L.load()
    heads = [_nhwc(h) for h in heads]
    B, Cp = heads[0].shape[:2]
    n = len(heads)
    L.check(lib.loft_rpn_decode_levels(L.arr(c_void_p, [h.data_ptr() for h in heads]), L.arr(c_void_p, [b.data_ptr() for b in base_anchors]),
                                       L.arr(c_int, [int(h.shape[2]) for h in heads]), L.arr(c_int, [int(h.shape[3]) for h in heads]),
                                       L.arr(c_int, [int(t) for t in topk]), L.arr(c_int, [int(s) for s in strides]),
                                       L.arr(c_int64, [int(o) for o in lvl_offs[:n]]), L.arr(c_int64, [int(o) for o in cand_offs[:n]]), n,
                                       L.ptr(sorted_idx), L.ptr(sorted_keys), B, Cp, A, c_int64(img_stride),
                                       L.arr(c_float, list(means)), L.arr(c_float, list(stds)), c_float(wh_ratio_clip),
                                       c_float(max_shape[0]), c_float(max_shape[1]), c_int64(cand_stride), L.ptr(out_boxes),
                                       L.ptr(out_scores), L.ptr(img_max), L.stream()), 'loft_rpn_decode_levels')


def rpn_finalize(top_scores, top_idx, cand_boxes, B, seg_stride, post):
    """-> (props fp32 [B, post, 5], counts int64 [B]) from the post-NMS top-k (loft_rpn_finalize)."""
    lib = L.load()
    L.dev_check(top_scores, top_idx, cand_boxes)
    props = torch.empty(B, post, 5, dtype=torch.float32, device=top_scores.device)
    counts = torch.empty(B, dtype=torch.int64, device=top_scores.device)
    L.check(lib.loft_rpn_finalize(L.ptr(top_scores), L.ptr(top_idx), L.ptr(cand_boxes), int(B), c_int64(seg_stride), int(post),
                                  L.ptr(props), L.ptr(counts), L.stream()), 'loft_rpn_finalize')
    return props, counts


def foa_targets(pos_boxes, pos_gt_offsets, stds=(0.5, 0.5)):
    lib = L.load()
    L.dev_check(pos_boxes, pos_gt_offsets)
    pos_boxes, pos_gt_offsets = pos_boxes.float().contiguous(), pos_gt_offsets.float().contiguous()
    n = pos_boxes.shape[0]
    out = torch.empty(4 * n, 2, dtype=torch.float32, device=pos_boxes.device)
    L.check(lib.loft_foa_targets(L.ptr(pos_boxes), L.ptr(pos_gt_offsets), c_int64(n), c_float(stds[0]), c_float(stds[1]),
                                 L.ptr(out), L.stream()), 'loft_foa_targets')
    return out


def foa_fuse_decode(pred, boxes, stds=(0.5, 0.5), max_shape=(1024, 1024)):
    lib = L.load()
    L.dev_check(pred, boxes)
    pred, boxes = pred.float().contiguous(), boxes[:, :4].float().contiguous()
    n = boxes.shape[0]
    out = torch.empty(n, 2, dtype=torch.float32, device=pred.device)
    L.check(lib.loft_foa_fuse_decode(L.ptr(pred), L.ptr(boxes), c_int64(n), c_float(stds[0]), c_float(stds[1]),
                                     c_float(max_shape[0]), c_float(max_shape[1]), L.ptr(out), L.stream()),
            'loft_foa_fuse_decode')
    return out


def offset_targets(pos_boxes, pos_gt_offsets, means=(0., 0.), stds=(0.5, 0.5), reg_num=2):
    """OffsetHead.get_targets (attribute_heads/offset_head.py:118-188): [n,4] boxes, [n,2] gt offsets -> [n,reg_num]."""
    lib = L.load()
    L.dev_check(pos_boxes, pos_gt_offsets)
    pos_boxes, pos_gt_offsets = pos_boxes.float().contiguous(), pos_gt_offsets.float().contiguous()
    n = pos_boxes.shape[0]
    out = torch.empty(n, reg_num, dtype=torch.float32, device=pos_boxes.device)
    L.check(lib.loft_offset_targets(L.ptr(pos_boxes), L.ptr(pos_gt_offsets), c_int64(n), c_float(means[0]), c_float(means[1]),
                                    c_float(stds[0]), c_float(stds[1]), c_int(reg_num), L.ptr(out), L.stream()),
            'loft_offset_targets')
    return out


def offset_decode(pred, boxes, means=(0., 0.), stds=(0.5, 0.5), max_shape=(1024, 1024), polar=False):
    """OffsetHead.get_offsets (attribute_heads/offset_head.py:190-243): pred [n,2|3], boxes [n,>=4] -> [n,2] pixels."""
    lib = L.load()
    L.dev_check(pred, boxes)
    pred, boxes = pred.float().contiguous(), boxes[:, :4].float().contiguous()
    n, reg_num = boxes.shape[0], int(pred.shape[1])
    out = torch.empty(n, 2, dtype=torch.float32, device=pred.device)
    L.check(lib.loft_offset_decode(L.ptr(pred), L.ptr(boxes), c_int64(n), c_float(means[0]), c_float(means[1]),
                                   c_float(stds[0]), c_float(stds[1]), c_float(max_shape[0]), c_float(max_shape[1]),
                                   c_int(reg_num), c_int(1 if polar else 0), L.ptr(out), L.stream()), 'loft_offset_decode')
    return out


def poly2mask(instances, H, W, device='cuda'):
    """instances: list (one per instance) of lists of polygons, each polygon a flat [x0, y0, x1, y1, ...] sequence (the BONAI
    ``segmentation`` / ``[footprint_mask]`` fields) -> uint8 [K, H, W] device tensor.  LoadAnnotations._poly2mask
    (datasets/pipelines/loading.py:301-326) for every instance of a tile in ONE launch; only the vertices cross PCIe."""
    lib = L.load()
    pk = instances if isinstance(instances, PackedPolygons) else pack_polygons(instances)
    K_ = pk.n
    out = torch.empty(K_, H, W, dtype=torch.uint8, device=device)
    if K_ == 0:
        return out
    xy = torch.from_numpy(pk.xy).to(device)
    poff_t = torch.from_numpy(pk.poff).to(device)
    ioff_t = torch.from_numpy(pk.ioff).to(device)
    L.check(lib.loft_poly2mask(L.ptr(xy), L.ptr(poff_t), L.ptr(ioff_t), K_, H, W, int(pk.maxv), L.ptr(out), L.stream()), 'loft_poly2mask')
    return out


def poly2mask_device(xy, poff, ioff, n, H, W, maxv):
    """loft_poly2mask on tables that are already on the device (fp64 [V,2], int64 offsets): the loader uploads them together with
    the batch's other small arrays in one copy."""
    lib = L.load()
    L.dev_check(xy, poff, ioff)
    out = torch.empty(n, H, W, dtype=torch.uint8, device=xy.device)
    if n:
        L.check(lib.loft_poly2mask(L.ptr(xy), L.ptr(poff), L.ptr(ioff), n, H, W, int(maxv), L.ptr(out), L.stream()), 'loft_poly2mask')
    return out


class PackedPolygons:
    """The host arrays loft_poly2mask reads (vertices fp64 [V,2], polygon offsets, instance offsets): built once per annotation
    (pack_polygons) and reusable -- the loader caches it per image instead of re-walking ~100 python lists every epoch."""
    __slots__ = ('xy', 'poff', 'ioff', 'maxv', 'n')

    def __init__(self, xy, poff, ioff, maxv, n):
        self.xy, self.poff, self.ioff, self.maxv, self.n = xy, poff, ioff, maxv, n


def pack_polygons(instances):
    import numpy as np
    pts, poff, ioff = [], [0], [0]
    for inst in instances:
        for poly in inst:
            a = np.asarray(poly, dtype=np.float64).reshape(-1, 2)
            pts.append(a)
            poff.append(poff[-1] + a.shape[0])
        ioff.append(len(poff) - 1)
    maxv = max((b - a for a, b in zip(poff[:-1], poff[1:])), default=0)
    xy = np.ascontiguousarray(np.concatenate(pts, 0) if pts else np.zeros((0, 2)), dtype=np.float64)
    return PackedPolygons(xy, np.asarray(poff, np.int64), np.asarray(ioff, np.int64), int(maxv), len(instances))


def mask_target(masks_u8, boxes, gt_idx, S=28):
    """masks: uint8 [K,H,W] (device), or a LIST of per-image uint8 [K_b,H,W] tensors (gt_idx then indexes their
    concatenation; no copy is made: the kernel gets a table of instance addresses); boxes [n,4] already clipped to the
    image, gt_idx int64 [n]."""
    lib = L.load()
    boxes = boxes.float().contiguous()
    n = boxes.shape[0]
    gt_idx = gt_idx.long().contiguous()
    out = torch.empty(n, S, S, dtype=torch.float32, device=boxes.device)
    if isinstance(masks_u8, (list, tuple)):
        ms = [m.contiguous() for m in masks_u8]
        L.dev_check(boxes, gt_idx, *ms)
        H, W = ms[0].shape[1], ms[0].shape[2]
        if any(m.dtype != torch.uint8 or tuple(m.shape[1:]) != (H, W) for m in ms):
            raise L.LoftHipError('mask_target: per-image masks must be uint8 [K,H,W] of one size')
        # (one vectorised range per image: a Python loop over the ~640 instances of a batch was ~100 us of host time in
        #  front of the mask branch)
        addr = h2d(np.concatenate([m.data_ptr() + np.arange(m.shape[0], dtype=np.int64) * (H * W) for m in ms])
                   if ms else np.zeros(0, np.int64), torch.int64, boxes.device)
        L.check(lib.loft_mask_target(None, H, W, L.ptr(boxes), L.ptr(gt_idx), c_int64(n), S, L.ptr(out), L.ptr(addr), L.stream()),
                'loft_mask_target')
        out._keep = ms      # the address table points into these tensors
        return out
    L.dev_check(masks_u8, boxes, gt_idx)
    masks_u8 = masks_u8.contiguous()
    L.check(lib.loft_mask_target(L.ptr(masks_u8), masks_u8.shape[1], masks_u8.shape[2], L.ptr(boxes),
                                 L.ptr(gt_idx), c_int64(n), S, L.ptr(out), None, L.stream()),
            'loft_mask_target')
    return out


# ------------------------------------------------------------------ inference post-processing

def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear'):
    """mmcv.ops.soft_nms contract on the device: -> (dets [M,5], inds int64 [M]) in selection order."""
    lib = L.load()
    L.dev_check(boxes, scores)
    boxes, scores = boxes.float().contiguous(), scores.float().contiguous()
    n = boxes.shape[0]
    dets = torch.zeros(n, 5, dtype=torch.float32, device=boxes.device)
    inds = torch.zeros(n, dtype=torch.int64, device=boxes.device)
    if n == 0:
        return dets, inds
    ws = torch.empty(lib.loft_soft_nms_workspace_bytes(n), dtype=torch.uint8, device=boxes.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    m = {'naive': 0, 'linear': 1, 'gaussian': 2}[method]
    L.check(lib.loft_soft_nms(L.ptr(boxes), L.ptr(scores), c_int64(n), c_float(iou_threshold), c_float(sigma),
                              c_float(min_score), m, L.ptr(ws), L.ptr(dets), L.ptr(inds), L.ptr(cnt), L.stream()),
            'loft_soft_nms')
    k = int(cnt.item())
    return dets[:k], inds[:k]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms: shift by idx*(max_coord+1), dispatch on nms_cfg['type']."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    if class_agnostic or boxes.numel() == 0:
        shifted = boxes
    else:
        shifted = boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]
    typ = cfg.pop('type', 'nms')
    if typ == 'nms':
        dets, keep = nms(shifted, scores, cfg['iou_threshold'], predicate=cfg.get('predicate', 'device'))
    elif typ == 'soft_nms':
        dets, keep = soft_nms(shifted, scores, **cfg)
    else:
        raise L.LoftHipError(f'unknown nms type {typ}')
    return torch.cat([boxes[keep], dets[:, -1:]], dim=-1), keep


def mask_paste(logits, boxes, img_h, img_w, thr=0.5):
    """logits fp32 [N,S,S], boxes [N,4] -> uint8 [N,img_h,img_w]."""
    lib = L.load()
    L.dev_check(logits, boxes)
    logits, boxes = logits.float().contiguous(), boxes.float().contiguous()
    N, S = logits.shape[0], logits.shape[-1]
    out = torch.empty(N, img_h, img_w, dtype=torch.uint8, device=logits.device)
    L.check(lib.loft_mask_paste(L.ptr(logits), L.ptr(boxes), N, S, int(img_h), int(img_w), c_float(thr), L.ptr(out),
                                L.stream()), 'loft_mask_paste')
    return out


def mask_translate(masks, offsets):
    """Footprints from roofs: masks uint8/bool [N,H,W], offsets fp32 [N,2] = (dx, dy) footprint -> roof (the model's offset
    output) -> uint8 [N,H,W] with out[n, y, x] = masks[n, y + round(dy), x + round(dx)], zero outside the image."""
    lib = L.load()
    L.dev_check(masks, offsets)
    m = masks.to(torch.uint8).contiguous()
    o = offsets.float().contiguous()
    if m.dim() != 3 or o.shape != (m.shape[0], 2):
        raise L.LoftHipError(f'mask_translate: masks [N,H,W] and offsets [N,2] expected, got {tuple(m.shape)} and {tuple(o.shape)}')
    out = torch.empty_like(m)
    L.check(lib.loft_mask_translate(L.ptr(m), L.ptr(o), m.shape[0], m.shape[1], m.shape[2], L.ptr(out), L.stream()),
            'loft_mask_translate')
    return out


# ------------------------------------------------------------------ sparse RPN backward helpers

def _sparse_levels(maps):
    maps = [_nhwc(m) for m in maps]
    for m in maps:
        if m.dtype != L.act16() or m.shape[1] != maps[0].shape[1]:
            raise L.LoftHipError('sparse RPN rows: bf16 maps with one channel count')
    ptrs = (c_void_p * len(maps))(*[m.data_ptr() for m in maps])
    return maps, ptrs, L.arr(c_int, [m.shape[2] for m in maps]), L.arr(c_int, [m.shape[3] for m in maps])


def rpn_gather_rows(maps, rows, K=3):
    """maps: list of bf16 channels_last [B,C,H_l,W_l]; rows int32 [nsel,4] (b, level, y, x) -> bf16 [nsel, K*K*C]."""
    lib = L.load()
    maps, ptrs, H, W = _sparse_levels(maps)
    L.dev_check(rows, *maps)
    C = maps[0].shape[1]
    out = torch.empty(rows.shape[0], K * K * C, dtype=L.act16(), device=rows.device)
    L.check(lib.loft_rpn_gather_rows(ptrs, H, W, len(maps), L.ptr(rows), rows.shape[0], C, K, L.ptr(out), L.stream()),
            'loft_rpn_gather_rows')
    return out


def rpn_sparse_prep(g, slot, A, P, w_cls, w_reg, w_conv):
    """loft_rpn_sparse_prep: g fp32 [nsel,5], slot int64 [nsel] -> (g_rows [nsel,P], w_headT [C,P], wd [9C,C]) in the activation
    type: the output-gradient rows in the fused head's channel order and the two dgrad operands, one launch."""
    lib = L.load()
    L.dev_check(g, slot, w_cls, w_reg, w_conv)
    nsel, C = int(g.shape[0]), int(w_conv.shape[0])
    if tuple(g.shape) != (nsel, 5) or tuple(w_conv.shape) != (C, C, 3, 3) or w_cls.numel() != A * C or w_reg.numel() != 4 * A * C:
        raise L.LoftHipError('rpn_sparse_prep: operand shapes')
    g = g.float().contiguous()
    dev = g.device
    g_rows = torch.empty(nsel, P, dtype=L.act16(), device=dev)
    w_headT = torch.empty(C, P, dtype=L.act16(), device=dev)
    wd = torch.empty(9 * C, C, dtype=L.act16(), device=dev)
    L.check(lib.loft_rpn_sparse_prep(L.ptr(g), L.ptr(slot.contiguous()), nsel, int(A), int(P), C, L.ptr(w_cls.float().contiguous()),
                                     L.ptr(w_reg.float().contiguous()), L.ptr(w_conv.float().contiguous()), L.ptr(g_rows),
                                     L.ptr(w_headT), L.ptr(wd), L.stream()), 'loft_rpn_sparse_prep')
    return g_rows, w_headT, wd


def rpn_scatter_add_rows_(maps, rows, src, K=3):
    """maps[level][b, y+dy, x+dx, :] += src[nsel, K*K*C] (in place, packed bf16 atomics)."""
    lib = L.load()
    maps, ptrs, H, W = _sparse_levels(maps)
    L.dev_check(rows, src)
    C = maps[0].shape[1]
    if src.dtype != L.act16() or tuple(src.shape) != (rows.shape[0], K * K * C) or not src.is_contiguous():
        raise L.LoftHipError('rpn_scatter_add_rows_: src must be contiguous bf16 [nsel, K*K*C]')
    L.check(lib.loft_rpn_scatter_add_rows(ptrs, H, W, len(maps), L.ptr(rows), rows.shape[0], C, K, L.ptr(src), L.stream()),
            'loft_rpn_scatter_add_rows')
    return maps


# ------------------------------------------------------------------ HRNet / HRFPN resampling and fusion

def fuse_sum_relu(terms, shifts, relu=True):
    """out = relu(sum_j nearest_up(terms[j], 1 << shifts[j])); the shift-0 terms define the output shape."""
    lib = L.load()
    terms = [_nhwc(t) for t in terms]
    L.dev_check(*terms)
    ref = terms[list(shifts).index(0)]
    B, C, H, W = ref.shape
    for t, s in zip(terms, shifts):
        if t.dtype != ref.dtype or tuple(t.shape) != (B, C, H >> s, W >> s):
            raise L.LoftHipError(f'fuse term {tuple(t.shape)} does not match {tuple(ref.shape)} >> {s}')
    out = empty_nhwc(B, C, H, W, ref.dtype, ref.device)
    ptrs = (c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    L.check(lib.loft_fuse_sum_relu(ptrs, L.arr(c_int, list(shifts)), len(terms), L.ptr(out), L.dtype_code(ref), B, H, W, C,
                                   int(relu), L.stream()), 'loft_fuse_sum_relu')
    return out


def blocksum_masked(g, y, shift):
    """[B,C,H,W] -> [B,C,H>>shift,W>>shift]: sum of g * (y > 0) over each block (y None: no mask)."""
    lib = L.load()
    g = _nhwc(g)
    y = _nhwc(y) if y is not None else None
    L.dev_check(g, y)
    B, C, H, W = g.shape
    out = empty_nhwc(B, C, H >> shift, W >> shift, g.dtype, g.device)
    L.check(lib.loft_blocksum_masked(L.ptr(g), L.ptr(y), L.ptr(out), L.dtype_code(g), B, H >> shift, W >> shift, C, shift,
                                     L.stream()), 'loft_blocksum_masked')
    return out


def bilinear_up_slot_(src, dst, shift, coff):
    """dst[:, coff:coff+C] = bilinear_up(src, 1 << shift) (align_corners=False); dst [B,Ctot,h<<shift,w<<shift]."""
    lib = L.load()
    src, dst = _nhwc(src), _nhwc(dst)
    L.dev_check(src, dst)
    B, C, h, w = src.shape
    if dst.dtype != src.dtype or tuple(dst.shape[2:]) != (h << shift, w << shift):
        raise L.LoftHipError('bilinear_up_slot_: destination does not match')
    L.check(lib.loft_bilinear_up_slot(L.ptr(src), L.ptr(dst), L.dtype_code(src), B, h, w, C, shift, dst.shape[1], coff, 0,
                                      L.stream()), 'loft_bilinear_up_slot')
    return dst


def bilinear_up_slot_bwd(gdst, C, shift, coff):
    """gradient of the slotted tensor [B,Ctot,H,W] -> gradient of the small map [B,C,H>>shift,W>>shift]."""
    lib = L.load()
    gdst = _nhwc(gdst)
    B, Ctot, H, W = gdst.shape
    out = empty_nhwc(B, C, H >> shift, W >> shift, gdst.dtype, gdst.device)
    L.check(lib.loft_bilinear_up_slot(L.ptr(gdst), L.ptr(out), L.dtype_code(gdst), B, H >> shift, W >> shift, C, shift, Ctot,
                                      coff, 1, L.stream()), 'loft_bilinear_up_slot(bwd)')
    return out


def avgpool(x, shift):
    lib = L.load()
    x = _nhwc(x)
    L.dev_check(x)
    B, C, H, W = x.shape
    out = empty_nhwc(B, C, H >> shift, W >> shift, x.dtype, x.device)
    L.check(lib.loft_avgpool(L.ptr(x), L.ptr(out), L.dtype_code(x), B, H >> shift, W >> shift, C, shift, 0, 0, L.stream()),
            'loft_avgpool')
    return out


def avgpool_bwd(g, shift):
    lib = L.load()
    g = _nhwc(g)
    B, C, Ho, Wo = g.shape
    out = empty_nhwc(B, C, Ho << shift, Wo << shift, g.dtype, g.device)
    L.check(lib.loft_avgpool(L.ptr(g), L.ptr(out), L.dtype_code(g), B, Ho, Wo, C, shift, 1, 0, L.stream()), 'loft_avgpool(bwd)')
    return out


def stem3x3s2_bn_relu(img, w, scale, shift, out_dtype=None):
    """img fp32 NCHW [B,3,H,W], w fp32 [64,3,3,3] -> channels_last [B,64,H/2,W/2] = relu(conv * scale + shift)."""
    lib = L.load()
    L.dev_check(img, w, scale, shift)
    img = img.float().contiguous()
    w, scale, shift = w.float().contiguous(), scale.float().contiguous(), shift.float().contiguous()
    B, _, H, W = img.shape
    out_dtype = out_dtype or L.act16()
    out = empty_nhwc(B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1, out_dtype, img.device)
    L.check(lib.loft_stem3x3s2_bn_relu(L.ptr(img), L.ptr(w), L.ptr(scale), L.ptr(shift), L.ptr(out), L.dtype_code(out), B, H, W,
                                       L.stream()), 'loft_stem3x3s2_bn_relu')
    return out


def stem3x3s2_wgrad(img, g, y):
    """-> (dwp fp32 [9,64,3] gradient of the BN-folded weight in the tap packing, db fp32 [64])."""
    lib = L.load()
    img = img.float().contiguous()
    g, y = _nhwc(g), _nhwc(y)
    L.dev_check(img, g, y)
    B, _, H, W = img.shape
    dwp = torch.zeros(9, 64, 3, dtype=torch.float32, device=img.device)
    db = torch.zeros(64, dtype=torch.float32, device=img.device)
    L.check(lib.loft_stem3x3s2_wgrad(L.ptr(img), L.ptr(g), L.ptr(y), L.ptr(dwp), L.ptr(db), L.dtype_code(g), B, H, W, L.stream()),
            'loft_stem3x3s2_wgrad')
    return dwp, db


# ------------------------------------------------------------------ DCNv2 sampling

def mdcn_sample_fwd(x, om, kh, kw, stride=1, pad=0, dil=1, deform_groups=1):
    """x [B,C,IH,IW] channels_last (bf16 | fp32), om fp32 [B,OMC,OH,OW] channels_last = raw conv_offset output
    -> col [B, kh*kw*C, OH, OW] channels_last (memory [B,OH,OW,K,C]), dtype of x."""
    lib = L.load()
    L.dev_check(x, om)
    x, om = _nhwc(x), _nhwc(om)
    if om.dtype != torch.float32:
        raise L.LoftHipError('conv_offset output must be fp32')
    B, C, IH, IW = x.shape
    _, omc, OH, OW = om.shape
    col = empty_nhwc(B, kh * kw * C, OH, OW, x.dtype, x.device)
    L.check(lib.loft_mdcn_sample_fwd(L.ptr(x), L.ptr(om), L.ptr(col), L.dtype_code(x), B, IH, IW, C, OH, OW, kh, kw, stride,
                                     pad, dil, deform_groups, omc, L.stream()), 'loft_mdcn_sample_fwd')
    return col


def mdcn_sample_bwd(x, om, dcol, kh, kw, stride=1, pad=0, dil=1, deform_groups=1):
    """-> (dx fp32 [B,C,IH,IW] channels_last, dom fp32 like om)."""
    lib = L.load()
    L.dev_check(x, om, dcol)
    x, om, dcol = _nhwc(x), _nhwc(om), _nhwc(dcol)
    if dcol.dtype != x.dtype:
        raise L.LoftHipError('dcol must have the dtype of x')
    B, C, IH, IW = x.shape
    _, omc, OH, OW = om.shape
    dx = zeros_nhwc(B, C, IH, IW, torch.float32, x.device)
    dom = zeros_nhwc(B, omc, OH, OW, torch.float32, x.device)
    nws = lib.loft_mdcn_bwd_workspace_bytes(B, C, OH, OW, kh, kw, stride, dil, deform_groups, omc)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws > 0 else None
    L.check(lib.loft_mdcn_sample_bwd(L.ptr(x), L.ptr(om), L.ptr(dcol), L.ptr(dx), L.ptr(dom), L.dtype_code(x), B, IH, IW, C,
                                     OH, OW, kh, kw, stride, pad, dil, deform_groups, omc, L.ptr(ws), L.stream()),
            'loft_mdcn_sample_bwd')
    return dx, dom


# ------------------------------------------------------------------ weight fold + pack

def fold_pack(w, conv_bias=None, bn=None, eps=1e-5, want_fwd=True, want_dgrad=True, out_fwd=None, out_dgrad=None,
              out_bias=None, dtype=None, cout_pad=None, cin_pad=None):
    """w fp32 [Cout,Cin,R,S]; bn = (gamma, beta, mean, var) or None -> (wp_fwd bf16 [T,Cout,Cin] | None,
    wp_dgrad bf16 [T,Cin,Cout] | None, bias fp32 [Cout])."""
    lib = L.load()
    L.dev_check(w, conv_bias)
    dtype = dtype or L.act16()
    w = w.contiguous()
    Cout, Cin, R, S = w.shape
    T = R * S
    dev = w.device
    CoutP, CinP = cout_pad or Cout, cin_pad or Cin       # channel-padded packings (zeros in the padding)
    wp = out_fwd if out_fwd is not None else (torch.empty(T, CoutP, CinP, dtype=dtype, device=dev) if want_fwd else None)
    wpt = out_dgrad if out_dgrad is not None else (torch.empty(T, CinP, CoutP, dtype=dtype, device=dev) if want_dgrad else None)
    bias = out_bias if out_bias is not None else torch.empty(CoutP, dtype=torch.float32, device=dev)
    g = b = m = v = None
    if bn is not None:
        g, b, m, v = [t.contiguous() for t in bn]
    cb = conv_bias.contiguous() if conv_bias is not None else None
    L.check(lib.loft_fold_pack(L.ptr(w), L.ptr(cb), L.ptr(g), L.ptr(b), L.ptr(m), L.ptr(v), c_float(eps), Cout, Cin, T,
                               L.ptr(wp), L.ptr(wpt), L.ptr(bias), int(dtype == torch.float32), CoutP, CinP, L.stream()),
            'loft_fold_pack')
    return wp, wpt, bias


FOLD_CHUNK = 2048
FOLD_TILE_MAX_RS = 9        # mirrors elementwise.hip
FOLD_NT_TAPS = 32            # output channels per tile of a record with 1 < RS <= FOLD_TILE_MAX_RS (elementwise.hip)


class PrepackRegistry:
    """All BN-fold + operand-packing jobs of a model as ONE launch per step (loft_fold_pack_multi).

    Jobs register themselves the first time a conv runs under the trainer (request()); from the next step on the trainer calls
    run() once at the start of the step -- the weights only change in the SGD kernel -- and request() just returns the buffers.
    Keys are the parameter storage addresses (stable: parameters live in the trainer's flat arena)."""

    def __init__(self):
        self.jobs, self.order = {}, []
        self.desc = None
        self.nchunks = 0
        self.step = -1
        # K8 re-arrangements of registry packings (loft_pack_k8_multi; the weight operands of loft_bneck_pair_bf16): address of the
        # packing -> dict(src, dst, step); one launch for all of them behind the batched packing launch
        self.k8jobs, self.k8desc = {}, None
        # fp32 parity mode (request_f32): fp32 packings + their binary16 planes for every registered conv, two launches per step
        self.f32jobs, self.f32order, self.f32desc = {}, [], None
        self.wplanes = {}            # (address, numel) of an fp32 packing -> (planes, absmax slot): kernels.WEIGHT_PLANES during a step

    def request(self, ws, conv_biases, bn, eps, cout_p, cin_p, want_dgrad, flat_chw=None):
        """ws / conv_biases: the G parameters of a grouped launch (G = 1 for a plain conv).
        -> (wp [G,T,CoutP,CinP], wpt [G,T,CinP,CoutP] | None, bias [G,CoutP]) bf16 packings valid for the current step.
        flat_chw = (C, H, W): ws[0] is a Linear weight [O, C*H*W] applied to a flattened map that is kept NHWC in memory --
        the packings come back as the 1x1 operands [1,1,O,H*W*C] / [1,1,H*W*C,O] with the K axis in (h, w, c) order."""
        key = (tuple(w.data_ptr() for w in ws), tuple(0 if b is None else b.data_ptr() for b in conv_biases),
               0 if bn is None else (bn[0].data_ptr(), bn[2].data_ptr()), cout_p, cin_p, bool(want_dgrad), flat_chw)
        grp = self.jobs.get(key)
        if grp is None:
            G = len(ws)
            dev = ws[0].device
            if flat_chw is not None:
                C, H, W = flat_chw
                Cout, Cin, RS, T, kin = ws[0].shape[0], C, -(H * W), 1, C * H * W      # RS < 0: n-major packing
                cout_p, cin_p = Cout, C
            else:
                Cout, Cin = ws[0].shape[0], ws[0].shape[1]
                RS = T = (ws[0].shape[2] * ws[0].shape[3]) if ws[0].dim() == 4 else 1
                kin = cin_p
            grp = dict(wp=torch.empty(G, T, cout_p, kin, dtype=L.act16(), device=dev),
                       wpt=torch.empty(G, T, kin, cout_p, dtype=L.act16(), device=dev) if want_dgrad else None,
                       bias=torch.empty(G, cout_p, dtype=torch.float32, device=dev), step=-2, members=[], flat=flat_chw)
            for g in range(G):
                grp['members'].append(dict(w=ws[g], cb=conv_biases[g], bn=bn, eps=float(eps), dims=(Cout, Cin, RS, cout_p, cin_p),
                                           wp=grp['wp'][g], wpt=None if grp['wpt'] is None else grp['wpt'][g], bias=grp['bias'][g]))
            self.jobs[key] = grp
            self.order.append(key)
            self.desc = None
        if grp['step'] != self.step:          # registered after this step's batched launch (first step): pack it now
            for j in grp['members']:
                w4 = j['w']
                if grp['flat'] is not None:   # (once: the permuted copy the batched launch avoids from the next step on)
                    C, H, W = grp['flat']
                    w4 = w4.detach().view(-1, C, H, W).permute(0, 2, 3, 1).reshape(w4.shape[0], -1)
                if w4.dim() == 2:
                    w4 = w4.detach().view(w4.shape[0], w4.shape[1], 1, 1)
                cop, cip = (j['dims'][3], j['dims'][4]) if grp['flat'] is None else (w4.shape[0], w4.shape[1])
                fold_pack(w4, j['cb'], j['bn'], j['eps'], want_dgrad=j['wpt'] is not None, out_fwd=j['wp'], out_dgrad=j['wpt'],
                          out_bias=j['bias'], cout_pad=cop, cin_pad=cip)
            grp['step'] = self.step
        return grp['wp'], grp['wpt'], grp['bias']

    def request_f32(self, ws, conv_biases, bn, eps, cout_p, cin_p, want_dgrad):
        """The fp32 parity mode's form of request(): fp32 packings (wp [G,T,CoutP,CinP], wpt [G,T,CinP,CoutP] | None, bias [G,CoutP])
        valid for the current step, folded by ONE launch for all registered convs (loft_fold_f32_multi), and -- where the plane
        kernels serve the shape -- their binary16 planes by a second one (loft_split_planes_f32_multi), which split_planes() then
        finds in self.wplanes instead of running an absmax + split launch per packing and step."""
        key = (tuple(w.data_ptr() for w in ws), tuple(0 if b is None else b.data_ptr() for b in conv_biases),
               0 if bn is None else (bn[0].data_ptr(), bn[2].data_ptr()), cout_p, cin_p, bool(want_dgrad))
        grp = self.f32jobs.get(key)
        if grp is None:
            G, dev = len(ws), ws[0].device
            Cout, Cin = ws[0].shape[0], ws[0].shape[1]
            T = (ws[0].shape[2] * ws[0].shape[3]) if ws[0].dim() == 4 else 1
            wp = torch.empty(G, T, cout_p, cin_p, dtype=torch.float32, device=dev)
            wpt = torch.empty(G, T, cin_p, cout_p, dtype=torch.float32, device=dev) if want_dgrad else None
            n = wp.numel()
            planes_ok = F32_CONTRACT in (F32_PLANES_F16, F32_PLANES_F16X4) and n % 8 == 0 and (T * cout_p * cin_p) % 8 == 0
            grp = dict(ws=ws, cbs=conv_biases, bn=bn, eps=float(eps), dims=(Cout, Cin, T, cout_p, cin_p), wp=wp, wpt=wpt,
                       bias=torch.empty(G, cout_p, dtype=torch.float32, device=dev), step=-2,
                       slot=self._f32_slot(dev),
                       # (forward launch: Cin % 64, Cout % 128; data-gradient launch: the roles swap)
                       pf=torch.empty(2, n, dtype=torch.float16, device=dev) if (planes_ok and cin_p % 64 == 0 and cout_p % 128 == 0) else None,
                       pd=torch.empty(2, n, dtype=torch.float16, device=dev)
                       if (planes_ok and want_dgrad and cout_p % 64 == 0 and cin_p % 128 == 0) else None)
            self.f32jobs[key] = grp
            self.f32order.append(key)
            self.f32desc = None
        if grp['step'] != self.step:          # registered after this step's batched launches (first step): run its records now
            self._run_f32([grp], None)
            grp['step'] = self.step
        if grp['pf'] is not None:
            self.wplanes[(grp['wp'].data_ptr(), grp['wp'].numel())] = (grp['pf'], grp['slot'])
        if grp['pd'] is not None:
            self.wplanes[(grp['wpt'].data_ptr(), grp['wpt'].numel())] = (grp['pd'], grp['slot'])
        return grp['wp'], grp['wpt'], grp['bias']

    def _f32_slot(self, dev):
        """Two words (absmax, spare) of the registry's slot arena: one memset per step for all records, and a record's slot never
        moves (a captured launch may hold its address)."""
        if getattr(self, 'f32slots', None) is None:
            self.f32slots, self.f32nslots = torch.zeros(4096, 2, dtype=torch.float32, device=dev), 0
        if self.f32nslots >= self.f32slots.shape[0]:
            raise L.LoftHipError('PrepackRegistry: more than 4096 fp32 records')
        self.f32nslots += 1
        return self.f32slots[self.f32nslots - 1]

    def _run_f32(self, groups, cached, zero=True):
        """The two launches for `groups`; cached = (fold desc, nblocks, n, split desc, nblocks, n, slots) of a previous call | None."""
        import struct
        plib = L.load_for(torch.float16)
        if cached is None:
            p = lambda t: 0 if t is None else t.data_ptr()
            frows, srows, fb, sb = [], [], 0, 0
            for grp in groups:
                Cout, Cin, T, CoutP, CinP = grp['dims']
                bn = grp['bn']
                eps_bits = struct.unpack('<i', struct.pack('<f', grp['eps']))[0]
                per = T * CoutP * CinP
                for g, w in enumerate(grp['ws']):
                    frows.append([p(w), p(grp['cbs'][g]), p(bn[0]) if bn else 0, p(bn[1]) if bn else 0, p(bn[2]) if bn else 0,
                                  p(bn[3]) if bn else 0, p(grp['wp'][g]), p(None if grp['wpt'] is None else grp['wpt'][g]), p(grp['bias'][g]),
                                  eps_bits, Cout, Cin, T, CoutP, CinP, fb, p(grp['slot']), 0])
                    fb += (per + 1023) // 1024
                    for src, pl in ((grp['wp'], grp['pf']), (grp['wpt'], grp['pd'])):
                        if pl is not None:
                            srows.append([src[g].data_ptr(), pl.data_ptr() + 2 * g * per, pl.shape[1], per, p(grp['slot']), sb])
                            sb += (per // 8 + 255) // 256
            dev = groups[0]['wp'].device
            cached = (h2d(frows, torch.int64, dev), fb, len(frows), h2d(srows, torch.int64, dev) if srows else None, sb, len(srows))
        fdesc, fb, fn, sdesc, sb, sn = cached
        if zero:
            for grp in groups:
                grp['slot'].zero_()
        L.check(plib.loft_fold_f32_multi(L.ptr(fdesc), fn, c_int64(fb), L.stream()), 'loft_fold_f32_multi')
        if sn:
            L.check(plib.loft_split_planes_f32_multi(L.ptr(sdesc), sn, c_int64(sb), L.stream()), 'loft_split_planes_f32_multi')
        return cached

    def _run_f32_cached(self, groups):
        plib = L.load_for(torch.float16)
        fdesc, fb, fn, sdesc, sb, sn = self.f32desc
        L.check(plib.loft_fold_f32_multi(L.ptr(fdesc), fn, c_int64(fb), L.stream()), 'loft_fold_f32_multi')
        if sn:
            L.check(plib.loft_split_planes_f32_multi(L.ptr(sdesc), sn, c_int64(sb), L.stream()), 'loft_split_planes_f32_multi')

    def k8(self, m):
        """K8 layout ([K/8, rows, 8]) of the registry packing m ([rows, K] view of a buffer request() returned), valid for the
        current step; None when m is not one of the registry's buffers (the caller converts it itself)."""
        job = self.k8jobs.get(m.data_ptr())
        if job is None:
            owned = any(m.data_ptr() == t.data_ptr() for grp in self.jobs.values() for t in (grp['wp'], grp['wpt']) if t is not None)
            if not owned or m.dim() != 2 or m.shape[1] % 8 or not m.is_contiguous():
                return None
            job = dict(src=m, dst=torch.empty(m.shape[1] // 8, m.shape[0], 8, dtype=m.dtype, device=m.device), step=-2)
            self.k8jobs[m.data_ptr()] = job
            self.k8desc = None
        if job['step'] != self.step:          # registered after this step's batched launch: convert it now
            pack_k8([job['src']], [job['dst']])
            job['step'] = self.step
        return job['dst']

    def run(self, step):
        """One launch for every registered job; afterwards request() is a dictionary lookup."""
        self.step = step
        if self.order:
            self._run_16(step)
        if self.f32order:
            groups = [self.f32jobs[k] for k in self.f32order]
            self.f32slots[:self.f32nslots].zero_()
            if self.f32desc is None:
                self.f32desc = self._run_f32(groups, None, zero=False)
            else:
                self._run_f32_cached(groups)
            for grp in groups:
                grp['step'] = step
        if self.k8jobs:
            js = list(self.k8jobs.values())
            _, self.k8desc = pack_k8([j['src'] for j in js], [j['dst'] for j in js], self.k8desc)
            for j in js:
                j['step'] = step

    def _run_16(self, step):
        lib = L.load()
        if self.desc is None:
            import struct
            rows, chunk = [], 0
            for j in (m for key in self.order for m in self.jobs[key]['members']):
                Cout, Cin, RS, CoutP, CinP = j['dims']
                bn = j['bn']
                p = lambda t: 0 if t is None else t.data_ptr()
                eps_bits = struct.unpack('<i', struct.pack('<f', j['eps']))[0]
                rows.append([p(j['w']), p(j['cb']), p(bn[0]) if bn else 0, p(bn[1]) if bn else 0, p(bn[2]) if bn else 0,
                             p(bn[3]) if bn else 0, p(j['wp']), p(j['wpt']), p(j['bias']), eps_bits, Cout, Cin, RS, CoutP, CinP, chunk])
                if RS < 0:
                    chunk += Cout             # n-major record: one output channel per chunk
                elif RS > FOLD_TILE_MAX_RS:
                    chunk += (CoutP * CinP * RS + FOLD_CHUNK - 1) // FOLD_CHUNK
                else:                         # one chunk per [NT output channels] x [64 input channels] tile, all taps
                    nt = 64 if RS == 1 else FOLD_NT_TAPS
                    chunk += ((CoutP + nt - 1) // nt) * ((CinP + 63) // 64)
            self.nchunks = chunk
            dev = self.jobs[self.order[0]]['wp'].device
            self.desc = h2d(rows, torch.int64, dev)
            self.nrows = len(rows)
        L.check(lib.loft_fold_pack_multi(L.ptr(self.desc), self.nrows, c_int64(self.nchunks), L.stream()), 'loft_fold_pack_multi')
        for key in self.order:                # n-major records: the [K][O] operand is the transpose of the [O][K] one
            grp = self.jobs[key]
            if grp['flat'] is not None and grp['wpt'] is not None:
                for j in grp['members']:
                    O, Kd = j['wp'].shape[-2], j['wp'].shape[-1]
                    L.check(lib.loft_transpose_bf16(L.ptr(j['wp']), L.ptr(j['wpt']), O, Kd, L.stream()), 'loft_transpose_bf16')
        for key in self.order:
            self.jobs[key]['step'] = step



_LOSS_SCRATCH = {}


def _loss_view(t, row_dims):
    """(d1, d2, s0, s1, s2) addressing the first `row_dims` dims of `t` in logical row-major order, or None if they do not
    collapse to three strided dims (loft_fused_loss_v2)."""
    dims = [(int(n), int(st)) for n, st in zip(t.shape[:row_dims], t.stride()[:row_dims]) if n != 1]
    merged = []
    for n, st in dims:
        if merged and merged[-1][1] == st * n:
            merged[-1] = (merged[-1][0] * n, st)
        else:
            merged.append((n, st))
    if len(merged) > 3:
        return None
    while len(merged) < 3:
        merged.insert(0, (1, 0))
    (_, s0), (d1, s1), (d2, s2) = merged
    if d1 == 1 and d2 == 1:
        return 1, 1, s2 if merged[2][0] != 1 or s2 else 1, 0, 0
    return d1, d2, s0, s1, s2


def fused_loss(mode, pred, target, weight=None, avg_factor=None, count=None, scale=1.0, beta=1.0, want_acc=False,
               target_ge1=False):
    """loft_fused_loss_v2: -> (loss fp32 [1] (+ top-1 accuracy in [1] with want_acc), grad fp32 dense like pred).
    mode: 'l1' | 'smooth_l1' | 'bce' | 'ce'.  pred may be a strided fp32 view (a column block / channel / row range of a wider head
    output): it is read in place.  weight: fp32 or bool, the shape of pred or its rows ('ce'), or a view expanded over pred's last
    dim (one weight per row, read un-expanded).  target_ge1 ('bce'): target are int64 labels, the target value is label >= 1.
    avg_factor: device scalar tensor, python number or None (then `count`, default = number of elements / rows)."""
    lib = L.load()
    L.dev_check(pred, target, weight)
    m = {'l1': 0, 'smooth_l1': 1, 'bce': 2, 'ce': 3}[mode]
    if pred.dtype != torch.float32:
        pred = pred.float()
    dev = pred.device
    if m == 3:
        n, C = pred.shape[0], pred.shape[1]
        if pred.stride(1) != 1 and C > 1:
            pred = pred.contiguous()
        view = _loss_view(pred, 1)
        target = target.contiguous()
        if target.dtype != torch.int64:
            target = target.long()
    else:
        n, C = pred.numel(), 1
        view = _loss_view(pred, pred.dim())
        if target_ge1:
            target = target.contiguous()
            if target.dtype != torch.int64:
                target = target.long()
        else:
            target = target.float().contiguous()
    if view is None:
        pred = pred.contiguous()
        view = (1, 1, C if m == 3 else 1, 0, 0)
    wdiv, wkind = 1, 0
    if weight is not None:
        if m != 3 and weight.dim() == pred.dim() and weight.dim() > 1 and weight.shape == pred.shape and weight.stride(-1) == 0:
            wdiv = int(pred.shape[-1])              # expanded over the last dim: one weight per row
            weight = weight[..., 0]
        elif m != 3 and weight.shape != pred.shape:
            weight = weight.expand(pred.shape)
        if weight.dtype in (torch.bool, torch.uint8):
            wkind = 1
            weight = weight.contiguous().view(torch.uint8)
        else:
            weight = weight.float().contiguous()
    af = None
    cnt = float(n if count is None else count)
    if torch.is_tensor(avg_factor):
        af = avg_factor.reshape(1)
        if af.dtype != torch.float32:
            af = af.float()
    elif avg_factor is not None:
        cnt = float(avg_factor)
    key = (dev.index, L.stream().value)
    if key not in _LOSS_SCRATCH:             # per stream: the ticket counter must not be shared by concurrent launches
        _LOSS_SCRATCH[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    counter = _LOSS_SCRATCH[key]
    partial = torch.empty(512, dtype=torch.float32, device=dev)
    grad = torch.empty(pred.shape, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    d1, d2, s0, s1, s2 = view
    L.check(lib.loft_fused_loss_v2(m, L.ptr(pred), c_int64(d1), c_int64(d2), c_int64(s0), c_int64(s1), c_int64(s2), L.ptr(target),
                                   1 if (target_ge1 and m == 2) else 0, L.ptr(weight), wkind, c_int64(wdiv), c_int64(n), int(C),
                                   L.ptr(af), c_float(max(cnt, 1e-30)), c_float(scale), c_float(beta), L.ptr(grad), L.ptr(partial),
                                   L.ptr(counter), L.ptr(out), 1 if (want_acc and m == 3) else 0, L.stream()), 'loft_fused_loss_v2')
    return (out[:1], grad, out[1:]) if want_acc else (out[:1], grad)


def rpn_sample_gather(heads, lvl_off, A, anchors, gts, gt_inds, pidx, pval, nidx, nval, means, stds):
    """Everything between the RPN's sampler and its losses in one launch (loft_rpn_sample_gather).
    heads: per level fp32 NHWC [B,Cp,H,W]; -> (vals [B,S,5], rows int32 [B*S,4], slot int64 [B*S], tgt [B,P,4],
    label int64 [B,S], weight fp32 [B,S])."""
    lib = L.load()
    L.dev_check(anchors, gts, gt_inds, pidx, nidx, *heads)
    heads = [_nhwc(h) for h in heads]
    B, Cp = heads[0].shape[0], heads[0].shape[1]
    P, Q = pidx.shape[1], nidx.shape[1]
    S = P + Q
    dev = gt_inds.device
    vals = torch.empty(B, S, 5, dtype=torch.float32, device=dev)
    rows = torch.empty(B * S, 4, dtype=torch.int32, device=dev)
    slot = torch.empty(B * S, dtype=torch.int64, device=dev)
    tgt = torch.empty(B, P, 4, dtype=torch.float32, device=dev)
    label = torch.empty(B, S, dtype=torch.int64, device=dev)
    weight = torch.empty(B, S, dtype=torch.float32, device=dev)
    pv, nv = pval.contiguous().view(torch.uint8), nval.contiguous().view(torch.uint8)
    Lv = len(heads)
    L.check(lib.loft_rpn_sample_gather(L.arr(c_void_p, [h.data_ptr() for h in heads]), L.arr(c_int, [h.shape[2] for h in heads]),
                                       L.arr(c_int, [h.shape[3] for h in heads]), L.arr(c_int64, list(lvl_off)), Lv, B, Cp, int(A),
                                       L.ptr(anchors.float().contiguous()), L.ptr(gts.float().contiguous()), int(gts.shape[1]),
                                       L.ptr(gt_inds.contiguous()), c_int64(gt_inds.shape[1]), L.ptr(pidx.contiguous()), L.ptr(pv), P,
                                       L.ptr(nidx.contiguous()), L.ptr(nv), Q, L.arr(c_float, list(means)), L.arr(c_float, list(stds)),
                                       L.ptr(vals), L.ptr(rows), L.ptr(slot), L.ptr(tgt), L.ptr(label), L.ptr(weight), L.stream()),
            'loft_rpn_sample_gather')
    return vals, rows, slot, tgt, label, weight


class _RoiTargets:
    """roi_sample_targets_begin's handle: ``rois_max`` (the first B * num_expected rows of the worst-case list; rows beyond the
    real count are zero boxes) can feed launches right away; ``finish()`` waits for the counts (the step's one host sync of the RoI head) and returns the
    exactly-sized views."""

    def __init__(self, out, counts_host, event, B, expected):
        self._out, self._counts, self._event, self._B = out, counts_host, event, B
        self.rois_max = out['rois'][:expected]      # the list a full sampler produces (num per image); zero boxes if it falls short

    def finish(self):
        self._event.synchronize()
        c = self._counts.tolist()
        B = self._B
        M = sum(c[0][:B]) + sum(c[1][:B])
        Np = sum(c[0][:B])
        o = self._out
        return dict(rois=o['rois'][:M], labels=o['labels'][:M], label_weights=o['label_weights'][:M],
                    bbox_targets=o['bbox_targets'][:M], bbox_weights=o['bbox_weights'][:M], pos_rois=o['pos_rois'][:Np],
                    pos_b=o['pos_b'][:Np], pos_gt_i=o['pos_gt_i'][:Np], pos_sel=o['pos_sel'][:Np])


def roi_sample_targets_begin(cand, gt_inds, gts, gt_labels, pidx, pval, nidx, nval, num_classes, means, stds, num_expected=None):
    """Sampled RoIs, labels and regression targets of a batch (loft_roi_sample_offsets + loft_roi_sample_targets), launched
    WITHOUT reading the per-image counts first: outputs are sized for the worst case (every sampler slot valid), the counts travel
    to pinned host memory behind the kernels.  -> _RoiTargets."""
    lib = L.load()
    L.dev_check(cand, gt_inds, gts, gt_labels, pidx, nidx, pval, nval)
    B, Ncand = gt_inds.shape
    P, Q = pidx.shape[1], nidx.shape[1]
    dev = gt_inds.device
    pv = pval.contiguous().view(torch.uint8) if pval.dtype == torch.bool else pval.contiguous()
    nv = nval.contiguous().view(torch.uint8) if nval.dtype == torch.bool else nval.contiguous()
    tab = torch.empty(4, B, dtype=torch.int32, device=dev)
    L.check(lib.loft_roi_sample_offsets(L.ptr(pv), L.ptr(nv), B, P, Q, L.ptr(tab), L.stream()), 'loft_roi_sample_offsets')
    Mx, Nx = B * (P + Q), B * P
    # (rois zero-filled: rows beyond the real count must be harmless boxes for a launch that runs before the count is known)
    out = dict(rois=torch.zeros(Mx, 5, device=dev), labels=torch.empty(Mx, dtype=torch.int64, device=dev),
               label_weights=torch.empty(Mx, device=dev), bbox_targets=torch.empty(Mx, 4, device=dev),
               bbox_weights=torch.empty(Mx, 4, device=dev), pos_rois=torch.empty(Nx, 5, device=dev),
               pos_b=torch.empty(Nx, dtype=torch.int64, device=dev), pos_gt_i=torch.empty(Nx, dtype=torch.int64, device=dev),
               pos_sel=torch.empty(Nx, dtype=torch.int64, device=dev))
    if Mx > 0:
        L.check(lib.loft_roi_sample_targets(L.ptr(cand.float().contiguous()), Ncand, L.ptr(gt_inds.contiguous()),
                                            L.ptr(gts.float().contiguous()), L.ptr(gt_labels.contiguous()), int(gts.shape[1]),
                                            L.ptr(pidx.contiguous()), L.ptr(nidx.contiguous()), P, Q, B, L.ptr(tab[0]), L.ptr(tab[1]),
                                            L.ptr(tab[2]), L.ptr(tab[3]), int(num_classes), L.arr(c_float, list(means)),
                                            L.arr(c_float, list(stds)), L.ptr(out['rois']), L.ptr(out['labels']),
                                            L.ptr(out['label_weights']), L.ptr(out['bbox_targets']), L.ptr(out['bbox_weights']),
                                            L.ptr(out['pos_rois']), L.ptr(out['pos_b']), L.ptr(out['pos_gt_i']), L.ptr(out['pos_sel']),
                                            L.stream()), 'loft_roi_sample_targets')
    host = torch.empty(2, B, dtype=torch.int32, pin_memory=True)
    host.copy_(tab[:2], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return _RoiTargets(out, host, ev, B, Mx if num_expected is None else min(Mx, B * int(num_expected)))


def roi_sample_targets(cand, gt_inds, gts, gt_labels, pidx, pval, nidx, nval, num_classes, means, stds):
    """-> dict(rois [M,5], labels, label_weights, bbox_targets, bbox_weights, pos_rois [Np,5], pos_b, pos_gt_i, pos_sel)."""
    return roi_sample_targets_begin(cand, gt_inds, gts, gt_labels, pidx, pval, nidx, nval, num_classes, means, stds).finish()


_SAMPLE_CALLS = [0]


def random_sample(gt_inds, num, max_pos, mode='random'):
    """RandomSampler.sample for a batch on the device (loft_random_sample).  gt_inds int64 [B,N] ->
    (pos_idx int64 [B,P], pos_valid bool [B,P], neg_idx int64 [B,Q], neg_valid bool [B,Q]), ascending indices per image.
    mode 'random': uniformly random subsets, deterministic in torch.initial_seed() and the call count; 'first': index order."""
    lib = L.load()
    L.dev_check(gt_inds)
    gt_inds = gt_inds.contiguous()
    if gt_inds.dtype != torch.int64:
        raise L.LoftHipError(f'gt_inds must be int64, got {gt_inds.dtype}')
    B, N = gt_inds.shape
    P, Q = min(int(max_pos), N), min(int(num), N)
    dev = gt_inds.device
    pidx = torch.empty(B, P, dtype=torch.int64, device=dev)
    nidx = torch.empty(B, Q, dtype=torch.int64, device=dev)
    pval = torch.empty(B, P, dtype=torch.uint8, device=dev)
    nval = torch.empty(B, Q, dtype=torch.uint8, device=dev)
    _SAMPLE_CALLS[0] += 1
    seed = (torch.initial_seed() * 6364136223846793005 + _SAMPLE_CALLS[0] * 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    if B > 0 and N > 0:
        ws = torch.empty(B * ((N + 15) // 16 * 16), dtype=torch.uint8, device=dev)
        L.check(lib.loft_random_sample(L.ptr(gt_inds), B, N, int(num), int(max_pos), 0 if mode == 'first' else 1,
                                       ctypes.c_uint64(seed), L.ptr(pidx), L.ptr(pval), L.ptr(nidx), L.ptr(nval), L.ptr(ws),
                                       L.stream()), 'loft_random_sample')
    return pidx, pval.view(torch.bool), nidx, nval.view(torch.bool)       # (the kernel writes 0 / 1 bytes: bool views, no copies)


def sampled_avg_factor(pos_valid, neg_valid):
    """sum_b max(#pos_b, 1) + sum_b max(#neg_b, 1) as a device fp32 scalar [1] (anchor_head.py:363-364), one launch."""
    lib = L.load()
    L.dev_check(pos_valid, neg_valid)
    pv, nv = pos_valid.contiguous().view(torch.uint8), neg_valid.contiguous().view(torch.uint8)
    out = torch.empty(1, dtype=torch.float32, device=pv.device)
    L.check(lib.loft_sampled_avg_factor(L.ptr(pv), L.ptr(nv), int(pv.shape[0]), int(pv.shape[1]), int(nv.shape[1]), L.ptr(out),
                                        L.stream()), 'loft_sampled_avg_factor')
    return out


def narrow_head_bwd(g, x, w, relu_in=False, need_gx=True, need_dw=True, need_db=True):
    """Backward of a 1x1 head with Cout <= 8 outputs in one pass over x (loft_narrow_head_bwd).
    g fp32 [N,c4,H,W] channels_last (c4 >= Cout), x bf16 NHWC [N,Cin,H,W], w fp32 [Cout,Cin]
    -> (gx bf16 NHWC | None, dw fp32 [Cout,Cin] | None, db fp32 [Cout] | None)."""
    lib = L.load()
    L.dev_check(g, x, w)
    x = _nhwc(x)
    g = g.float().contiguous(memory_format=torch.channels_last)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    w2 = w.reshape(Cout, Cin).float().contiguous()
    f32 = x.dtype == torch.float32              # fp32 parity mode (loft_narrow_head_bwd_f32, round 6)
    gx = empty_nhwc(N, Cin, H, W, torch.float32 if f32 else L.act16(), x.device) if need_gx else None
    dw = pooled_zeros((Cout, Cin), x.device) if need_dw else None
    db = pooled_zeros((Cout,), x.device) if need_db else None
    fn = lib.loft_narrow_head_bwd_f32 if f32 else lib.loft_narrow_head_bwd
    L.check(fn(L.ptr(g), int(g.shape[1]), L.ptr(x), L.ptr(w2), c_int64(N * H * W), Cin, Cout, int(relu_in),
               L.ptr(gx), L.ptr(dw), L.ptr(db), L.stream()), 'loft_narrow_head_bwd')
    return gx, dw, db


class UnpackQueue:
    """Deferred loft_fold_unpack_bwd jobs of the trainer's direct gradient sink, flushed as ONE launch
    (loft_fold_unpack_bwd_multi) every ``limit`` jobs and at the end of the backward pass."""

    def __init__(self, limit=48, note=None, stream=None):
        self.limit = limit
        self.note = note       # callable(params) -> bool: "flush now" (the reducer: a gradient bucket is complete in the queue)
        self.jobs, self.done = [], []
        # jobs may be produced on a side stream (the mask branch runs beside the FOA branch); the batched launch always goes to
        # the queue's HOME stream and waits for an event per foreign job.  stream=None: home = the stream the queue was created
        # on.  stream=<a dedicated stream> (the Trainer, round 6): the descriptor upload + unpack launch of a flush no longer sit
        # in the main stream between two data-gradient kernels -- nothing on the main stream reads the arena before the end of
        # the backward pass -- which is what made the reducer's early flushes (one per completed bucket) cost device time:
        # jobs produced on the creation ("main") stream are covered by ONE event recorded there at flush time, and the job
        # tensors are kept referenced until the trainer has joined the two streams (no block is handed back to the main
        # stream's allocator pool while the unpack stream may still read it).
        self.main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        self.main_raw = L.stream().value if self.main is not None else None
        self.home = stream if stream is not None else self.main
        self.home_raw = self.home.cuda_stream if stream is not None else self.main_raw
        self.dedicated = stream is not None
        self.events = []
        self.main_jobs = False
        self.keep = []
        self._targets = set()

    def add(self, dwp, db, w, bn, eps, slots, on_done=(), flat_chw=None, nsplit=1, params=()):
        """slots = (dw, dgamma | None, dbeta-or-dbias | None) arena views to accumulate into; on_done: callables run after the
        launch that served this job has been enqueued (the reducer's gradient-ready notifications).
        flat_chw = (C, H, W): w is a Linear weight [O, C*H*W] and dwp [O, H*W*C] its gradient in NHWC-flattened K order."""
        # One launch must not hold two records that accumulate into the SAME arena slot (a parameter used several times per step:
        # the RPN head's convs over five pyramid levels in the fp32 parity mode, where the sparse RPN backward does not apply): the
        # records of a batch run as concurrent workgroups and their read-modify-write of the slot is not atomic.  Found in round 6
        # (tests/test_trainer_gpu.py::test_fp32_mode_trainer_...: rpn_conv's gradient off by 8-30 %, differently every run).
        tgt = slots[0].data_ptr() if slots[0] is not None else None
        if tgt is not None:
            if tgt in self._targets:
                if os.environ.get('LOFT_DEBUG_TARGETS'):
                    print('unpack queue: slot used twice in one batch ->', tuple(w.shape), 'jobs pending', len(self.jobs), flush=True)
                self.flush()
            self._targets.add(tgt)
        self.jobs.append((dwp, db, w, bn, float(eps), slots, flat_chw, int(nsplit)))     # nsplit > 1: dwp = [nsplit][...] split-K slots
        if self.home is not None:
            raw = L.stream().value
            if raw == self.home_raw:
                pass
            elif self.dedicated and raw == self.main_raw:
                self.main_jobs = True             # (one event on the main stream at flush time covers all of these)
            else:
                ev = torch.cuda.Event()
                ev.record()                       # (on the producing side stream)
                self.events.append(ev)
                if not self.dedicated:
                    for t in (dwp, db):           # allocated on the side stream, read by the batched launch on the home stream
                        if t is not None:
                            t.record_stream(self.home)
            if self.dedicated:
                self.keep.append((dwp, db))
        self.done.extend(on_done)
        if len(self.jobs) >= self.limit or (self.note is not None and self.note(params)):
            self.flush()

    def join(self, stream=None):
        """Make ``stream`` (default: the creation stream) wait for every launch of the queue; drops the kept job tensors."""
        if self.dedicated:
            (stream or self.main).wait_stream(self.home)
        self.keep = []

    def flush(self):
        if self.home is not None and L.stream().value != self.home_raw:
            if self.main_jobs:
                ev = torch.cuda.Event()
                ev.record(self.main)              # everything the main stream has been handed so far (host order)
                self.events.append(ev)
                self.main_jobs = False
            with torch.cuda.stream(self.home):
                return self.flush()
        for ev in self.events:
            self.home.wait_event(ev)
        self.events = []
        if self.jobs:
            import struct
            rows, blk = [], 0
            p = lambda t: 0 if t is None else t.data_ptr()
            for dwp, db, w, bn, eps, (dw, dg, dbeta), flat, nsplit in self.jobs:
                if flat is not None:
                    Cout, Cin, RS = w.shape[0], flat[0], -(flat[1] * flat[2])
                    coutp, cinp = Cout, Cin
                else:
                    Cout, Cin = w.shape[0], w.shape[1]
                    RS = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
                    coutp, cinp = dwp.shape[-2], dwp.shape[-1]
                g, _, m, v = bn if bn is not None else (None, None, None, None)
                rows.append([p(dwp), p(db), p(w), p(g), p(m), p(v), p(dw), p(dg), p(dbeta),
                             (struct.unpack('<I', struct.pack('<f', eps))[0]) | (nsplit << 32), Cout, Cin, RS, coutp, cinp, blk])
                blk += Cout
            desc = h2d(rows, torch.int64, self.jobs[0][2].device)
            # dynamic LDS row: records that interleave taps through LDS (n-major Linear records must fit; conv records with
            # more than one tap use it when they fit, else the direct form)
            # (n-major records go through it in blocks of 64 channels: 64 * taps floats, not the whole 12 544-float row -- the
            #  dynamic LDS size is per LAUNCH, and 50 KB of it left three blocks per CU for every record of the batch)
            lds = max([min(r[11], 64) * abs(r[12]) if r[12] < 0 else r[11] * r[12] for r in rows
                       if r[12] < 0 or (r[12] > 1 and r[11] * r[12] <= 16384)], default=0)
            if lds > 16384:
                raise L.LoftHipError(f'fold_unpack_bwd_multi: a record needs {lds} floats of LDS')
            L.check(L.load().loft_fold_unpack_bwd_multi(L.ptr(desc), len(rows), c_int64(blk), int(lds), L.stream()),
                    'loft_fold_unpack_bwd_multi')
            self.jobs = []
            self._targets = set()
        done, self.done = self.done, []
        for f in done:
            f()


def fold_unpack_bwd(dwp, db, w, bn=None, eps=1e-5, need_dw=True, out=None):
    """dwp fp32 [T,CoutP,CinP], db fp32 [CoutP] | None -> (dw [Cout,Cin,R,S] | None, dgamma | None, dbeta | None).
    out = (dw, dgamma, dbeta) existing fp32 contiguous tensors (e.g. slots of a flat gradient arena): ACCUMULATE into them."""
    lib = L.load()
    w = w.contiguous()
    Cout, Cin, R, S = w.shape
    dev = w.device
    g = m = v = dg = dbeta = None
    if out is not None:
        dw, dg, dbeta = out
        if bn is not None:
            g, _, m, v = [t.contiguous() for t in bn]
    else:
        dw = torch.empty_like(w) if need_dw else None
        if bn is not None:
            g, _, m, v = [t.contiguous() for t in bn]
            dg = torch.empty(Cout, dtype=torch.float32, device=dev)
            dbeta = torch.empty(Cout, dtype=torch.float32, device=dev)
    L.check(lib.loft_fold_unpack_bwd(L.ptr(dwp), L.ptr(db), L.ptr(w), L.ptr(g), L.ptr(m), L.ptr(v), c_float(eps), Cout, Cin,
                                     R * S, L.ptr(dw), L.ptr(dg), L.ptr(dbeta), dwp.shape[-2], dwp.shape[-1], int(out is not None),
                                     L.stream()),
            'loft_fold_unpack_bwd')
    return dw, dg, dbeta
