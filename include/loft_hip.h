/*
 * loft_hip.h -- C-ABI of libloft_hip.so, the MI355X (gfx950) kernels of the LOFT/FOA hot path.
 *
 * The reference (jwwangchn/BONAI, a fork of MMDetection 2.3.0) has NO native code of its own:
 * mmdet/ops/__init__.py:1-32 re-exports mmcv.ops, and every kernel on the LOFT path lives in the
 * pinned third-party mmcv==1.0.5 or in torch/cuDNN.  This header is therefore the boundary a
 * maintainer binds where the reference today calls `mmcv.ops.*` / `torch.nn.functional.*`; each
 * entry cites the reference call site it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary.
 *   - every pointer is DEVICE memory unless the name ends in _host; the caller owns all buffers,
 *     kernels never allocate; workspaces are passed in.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - return value: 0 on success, otherwise a hipError_t code.  No exceptions cross the boundary.
 *   - re-entrant, no global state; one host thread per process (one process per GPU).
 *   - dtype codes: LOFT_F32 = 0, LOFT_BF16 = 1 (raw bfloat16 bits as uint16_t).
 *   - activations are NHWC ("channels_last"): [N][H][W][C], C contiguous.
 */
#ifndef LOFT_HIP_H
#define LOFT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOFT_F32 0
#define LOFT_BF16 1

/* ---- RoIAlign ---------------------------------------------------------------------------
 * Replaces SingleRoIExtractor.forward's per-level loop over mmcv.ops.RoIAlign
 * (mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:53-80; layers built at
 * base_roi_extractor.py:32-55 with sampling_ratio=0, aligned=True) and its backward.
 * feats[l]: NHWC feature map of level l ([B,H[l],W[l],C], dtype), scales[l] = 1/stride.
 * rois: [K,5] fp32 (batch_idx, x1, y1, x2, y2).  The level of each RoI is computed in-kernel
 * (map_roi_levels, :32-51).  out: [n_rot, K, P, P, C] in `dtype`; n_rot = 1, or 4 to emit the
 * four FOA rotations (offset_head_expand_feature.py:163-196) in the same pass.
 * bwd: grad_out has the layout of out; grad_feats[l] are fp32 NHWC accumulators (atomic adds;
 * the caller zeroes them).  H/W/scales are HOST arrays of num_levels entries. */
int loft_roi_align_fwd(const void* const* feats_host, const int* H_host, const int* W_host, const float* scales_host,
                       int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                       int n_rot, void* out, void* stream);
int loft_roi_align_bwd(float* const* grad_feats_host, const int* H_host, const int* W_host, const float* scales_host,
                       int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                       int n_rot, const void* grad_out, void* stream);
/* map_roi_levels alone (single_level_roi_extractor.py:32-51) -> int32 [K]. */
int loft_map_roi_levels(const float* rois, int K, int num_levels, int finest_scale, int32_t* out, void* stream);

/* ---- NMS --------------------------------------------------------------------------------
 * Replaces mmcv.ops.batched_nms -> nms (mmdet/models/dense_heads/rpn_head.py:166-168,
 * mmdet/core/post_processing/bbox_nms.py:63).  boxes [total,4] fp32 are already sorted by
 * (score desc, index asc) inside each segment; segment s spans rows
 * [seg_offsets[s], seg_offsets[s+1]).  seg_shift[s] (may be NULL) is added to all four
 * coordinates before the IoU test = batched_nms's idx*(max_coord+1) shift.  keep[total] gets
 * 1 for survivors.  Suppression: IoU > iou_thr, offset 0.  workspace: loft_nms_workspace_bytes. */
int64_t loft_nms_workspace_bytes(int64_t total_boxes, int64_t max_segment);
int loft_nms_segmented(const float* boxes, const int64_t* seg_offsets, const float* seg_shift, int num_segments,
                       int64_t total_boxes, int64_t max_segment, float iou_thr, void* workspace, uint8_t* keep,
                       void* stream);
/* Stable segmented sort by key, descending (ties keep input order) -- the `scores.sort(
 * descending=True)` of rpn_head.py:129 with a defined tie order.  Call with workspace == NULL to
 * query *workspace_bytes. */
int loft_segmented_sort_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                             int64_t num_items, int num_segments, const int64_t* seg_offsets, void* workspace,
                             int64_t* workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOFT_HIP_H */
