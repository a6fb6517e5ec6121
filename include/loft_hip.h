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
#define LOFT_F16 2
/* The library is built twice from the same sources: libloft_hip.so (16-bit type = bfloat16) and libloft_hip_f16.so (IEEE
 * binary16, for the reference's `fp16 = dict(loss_scale=512.)` configs, mmdet/core/fp16/hooks.py:11-135).  Both export this
 * same header; wherever a name or a comment below says "bf16" read "the 16-bit type of the library at hand", and wherever a
 * dtype code is taken the 16-bit code accepted is the library's own (LOFT_BF16 resp. LOFT_F16; the other one is rejected with
 * hipErrorInvalidValue).  MFMA: v_mfma_f32_32x32x16_bf16 resp. v_mfma_f32_32x32x16_f16 -- same rate, fp32 accumulation. */
int loft_act16_dtype(void);   /* LOFT_BF16 or LOFT_F16: which build this is */

/* ---- RoIAlign ---------------------------------------------------------------------------
 * Replaces SingleRoIExtractor.forward's per-level loop over mmcv.ops.RoIAlign
 * (mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:53-80; layers built at
 * base_roi_extractor.py:32-55 with sampling_ratio=0, aligned=True) and its backward.
 * feats[l]: NHWC feature map of level l ([B,H[l],W[l],C], dtype), scales[l] = 1/stride.
 * rois: [K,5] fp32 (batch_idx, x1, y1, x2, y2).  The level of each RoI is computed in-kernel
 * (map_roi_levels, :32-51).  out: [n_rot, K, P, P, C] in `dtype`; n_rot = 1, or 4 to emit the
 * four FOA rotations (offset_head_expand_feature.py:163-196) in the same pass.
 * bwd: grad_out has the layout of out; grad_feats[l] are fp32 NHWC maps [B,H[l],W[l],C], every pixel
 * of which is written (accumulate=0) or added to (accumulate=1) exactly once -- no atomics reach HBM.
 * rois_sorted=1 promises the RoIs are ordered by batch index (bbox2roi order) so each tile scans only its
 * image's RoIs; workspace: 48*K bytes (16-byte aligned).  H/W/scales are HOST arrays of num_levels entries.  grad_dtype: LOFT_F32, or LOFT_BF16 (only with dtype ==
 * LOFT_BF16): the gradient maps are written directly in bf16 (each pixel exactly once from fp32 registers). */
int loft_roi_align_fwd(const void* const* feats_host, const int* H_host, const int* W_host, const float* scales_host,
                       int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                       int n_rot, void* out, void* stream);
int loft_roi_align_bwd(void* const* grad_feats_host, const int* H_host, const int* W_host, const float* scales_host,
                       int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                       int n_rot, const void* grad_out, int B, int accumulate, int rois_sorted, void* workspace,
                       int grad_dtype, void* stream);
/* The same with the kernel chosen by the caller (tests / A-B timing; no environment variable is read anywhere in the library):
 * LOFT_ROI_AUTO = the shipped choice (16-bit forward: separable kernel; 16-bit maps with C == 256: per-(RoI, tile) GEMMs). */
#define LOFT_ROI_AUTO 0
#define LOFT_ROI_FWD_SAMPLE 1   /* forward: the sample-order kernel also for the 16-bit type */
#define LOFT_ROI_FWD_SEP4 2     /* forward: the separable kernel with 8-byte accesses (rounds 2-3; shipped only when C % 8 != 0) */
/* (forward, LOFT_ROI_AUTO only: bits 8-24 of `variant` tune the shipped kernel -- LDS stage KiB, smallest channel block, samples
 * per bin, workgroups per RoI; see loft_roi_align_fwd_ord in roi_align.hip.  0 = the shipped values.) */
#define LOFT_ROI_BWD_VALU 1     /* backward: the tile-owner VALU kernel also where the MFMA form applies */
#define LOFT_ROI_BWD_PIPE 2     /* backward: the per-pair GEMMs as a chunk pipeline (global -> LDS copies one chunk ahead; bit-identical maps, measured neutral: not shipped) */
int loft_roi_align_fwd_v(const void* const* feats_host, const int* H_host, const int* W_host, const float* scales_host,
                         int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                         int n_rot, void* out, int variant, void* stream);
/* Launch order for a RoI list (SingleRoIExtractor receives the sampler's list in sampling order,
 * single_level_roi_extractor.py:53-80): order[K] int32 (device) = the RoI indices bucketed by (image, level, row strip of the
 * level's map), one workgroup, LDS counting sort.  loft_roi_align_fwd_ord = loft_roi_align_fwd_v whose workgroups serve the
 * RoIs in that order, one contiguous eighth of it per XCD, so overlapping windows are fetched from HBM once per XCD instead
 * of once per RoI; order = NULL is the unordered launch.  Outputs are bit-identical either way (same arithmetic, same rows).
 * H_host / scales_host: per-level map heights and 1/stride; B = number of images (batch indices are clamped to it). */
int loft_roi_order(const int* H_host, const float* scales_host, int num_levels, int finest_scale, const float* rois, int K, int B,
                   int32_t* order, void* stream);
int loft_roi_align_fwd_ord(const void* const* feats_host, const int* H_host, const int* W_host, const float* scales_host,
                           int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                           int n_rot, void* out, int variant, const int32_t* order, void* stream);
int loft_roi_align_bwd_v(void* const* grad_feats_host, const int* H_host, const int* W_host, const float* scales_host,
                         int num_levels, int finest_scale, int C, int dtype, const float* rois, int K, int P,
                         int n_rot, const void* grad_out, int B, int accumulate, int rois_sorted, void* workspace,
                         int grad_dtype, int variant, void* stream);
/* nsets RoI lists over the SAME pyramid in one pass per level: the LOFT head runs three SingleRoIExtractors (bbox 7x7,
 * mask 14x14, FOA 14x14 x 4 rotations: loft_foa.py:126-176 calls bbox_roi_extractor / mask_roi_extractor / the offset
 * head's extractor on the same FPN maps), so their backward passes add into the same four maps; fused (16-bit maps,
 * C == 256, nsets <= 3) every map pixel is written once.  Per-list arguments are host arrays of length nsets;
 * workspace[i] holds 48*K[i] bytes.  Other configurations run list after list, accumulating. */
int loft_roi_align_bwd_multi(void* const* grad_feats_host, const int* H_host, const int* W_host, const float* scales_host,
                             int num_levels, int finest_scale, int C, int dtype, int nsets,
                             const float* const* rois_host, const int* K_host, const int* P_host, const int* n_rot_host,
                             const void* const* grad_out_host, int B, int accumulate, const int* rois_sorted_host,
                             void* const* workspace_host, int grad_dtype, void* stream);
/* map_roi_levels alone (single_level_roi_extractor.py:32-51) -> int32 [K]. */
int loft_map_roi_levels(const float* rois, int K, int num_levels, int finest_scale, int32_t* out, void* stream);

/* ---- NMS --------------------------------------------------------------------------------
 * Replaces mmcv.ops.batched_nms -> nms (mmdet/models/dense_heads/rpn_head.py:166-168,
 * mmdet/core/post_processing/bbox_nms.py:63).  boxes [total,4] fp32 are already sorted by
 * (score desc, index asc) inside each segment; segment s spans rows
 * [seg_offsets[s], seg_offsets[s+1]).  seg_shift[s] (may be NULL) is added to all four
 * coordinates before the IoU test = batched_nms's idx*(max_coord+1) shift.  keep[total] gets
 * 1 for survivors.  Suppression: IoU > iou_thr, offset 0.  workspace: loft_nms_workspace_bytes. */
int64_t loft_nms_workspace_bytes(int64_t total_boxes, int64_t max_segment, int64_t num_segments);
/* loft_nms_segmented_levels: segments = (image, level) pairs, image-major; batched_nms's shift idx * (boxes.max() + 1) is derived on
 * the device from img_max_dev [num_segments / levels] (loft_rpn_decode_levels) -- same fp32 operations as the tensor expression. */
int loft_nms_segmented_levels(const float* boxes, const int64_t* seg_offsets, const float* img_max_dev, int levels, int num_segments,
                              int64_t total_boxes, int64_t max_segment, float iou_thr, int predicate, void* workspace,
                              uint8_t* keep, void* stream);
int loft_nms_segmented(const float* boxes, const int64_t* seg_offsets, const float* seg_shift, int num_segments,
                       int64_t total_boxes, int64_t max_segment, float iou_thr, void* workspace, uint8_t* keep,
                       void* stream);
/* The same with the suppression predicate chosen by the caller.  mmcv-1.0.5 (not in the reference tree) has two:
 * LOFT_NMS_PRED_DEVICE  inter > thr * union    -- its CUDA kernel, i.e. what the reference's GPU training and
 *                                                 tools/test.py runs execute; loft_nms_segmented's behaviour, the default;
 * LOFT_NMS_PRED_CPU     inter / union >= thr   -- its host path (nms_cpu).
 * They differ exactly at IoU == thr (kept by the device form, suppressed by the host form). */
#define LOFT_NMS_PRED_DEVICE 0
#define LOFT_NMS_PRED_CPU 1
int loft_nms_segmented_pred(const float* boxes, const int64_t* seg_offsets, const float* seg_shift, int num_segments,
                            int64_t total_boxes, int64_t max_segment, float iou_thr, int predicate, void* workspace,
                            uint8_t* keep, void* stream);
/* Stable segmented sort by key, descending (ties keep input order) -- the `scores.sort(
 * descending=True)` of rpn_head.py:129 with a defined tie order.  Call with workspace == NULL to
 * query *workspace_bytes. */
int loft_segmented_sort_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                             int64_t num_items, int num_segments, const int64_t* seg_offsets, void* workspace,
                             int64_t* workspace_bytes, void* stream);
/* loft_segmented_topk_desc: the first k (<= 4096) entries of every segment's STABLE descending order -- all the training step reads
 * of `scores.sort(descending=True)[:nms_pre]` per (image, level) (rpn_head.py:129-136) and of the post-NMS `dets[:nms_post]`
 * (rpn_head.py:166-168).  In-house radix select + LDS bitonic sort, one workgroup per segment, no workspace.  Outputs have the
 * layout of loft_segmented_sort_desc (entry r of segment s at seg_offsets[s] + r); entries r >= min(k, segment length) are NOT
 * written.  vals_in may be NULL (values = global element indices).  out_offsets_dev (optional, device int64 [num_segments]): segment s
 * writes its head at out_offsets[s] instead of seg_offsets[s] (two-stage selection of long segments). */
int loft_segmented_topk_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out, int num_segments,
                             const int64_t* seg_offsets_dev, int k, const int64_t* out_offsets_dev, const uint8_t* key_mask_dev,
                             void* stream);
/* key_mask_dev (optional, one byte per key): keys whose byte is 0 count as -1.0 -- the NMS keep flags, i.e. rpn_head.py:169's
 * `dets = dets[keep]` expressed as a masked score without a pass of its own.
 * loft_topk_merge_runs: second stage of the selection for segments too long for one workgroup.  cand_keys / cand_vals hold, per
 * sub-range ("run") of a segment, that run's first <= k entries in stable descending order (loft_segmented_topk_desc with
 * out_offsets); an entry's rank in the merged order is found by one binary search per other run of its segment; entries with
 * rank < k land at run_out[r] + rank.  run_offsets_dev int64 [num_runs + 1], run_first_dev / run_count_dev int32 [num_runs] (the runs
 * of the run's segment), run_out_dev int64 [num_runs]; max_run = the longest run. */
int loft_topk_merge_runs(const float* cand_keys, const int32_t* cand_vals, int num_runs, int max_run, const int64_t* run_offsets_dev,
                         const int32_t* run_first_dev, const int32_t* run_count_dev, const int64_t* run_out_dev, int k,
                         float* keys_out, int32_t* vals_out, void* stream);

/* ---- dense contractions on MFMA -----------------------------------------------------------
 * loft_conv_tap_bf16: im2col-free NHWC convolution / linear layer, bf16 operands, fp32 accumulate.
 * Replaces the nn.Conv2d (cuDNN) / nn.Linear / nn.ConvTranspose2d calls of
 * mmdet/models/backbones/resnet.py:266-298, necks/fpn.py:170-199, dense_heads/rpn_head.py:38-44,
 * roi_heads/bbox_heads/convfc_bbox_head.py:135-173, roi_heads/mask_heads/fcn_mask_head.py:118-126,
 * roi_heads/attribute_heads/offset_head_expand_feature.py:134-161 -- forward AND data-gradient
 * (the caller supplies the tap table and the matching weight packing).
 *
 *   out[g][b, oy*os+oo_y, ox*os+oo_x, n] = act( bias[g][n] + residual[same index] +
 *        sum_{t<T} sum_{c<Cin} src[g][b, oy*ss+dy[t], ox*ss+dx[t], c] * wgt[g][wt[t]][n][c] )
 *
 * src [B,IH,IW,Cin] bf16; wgt [taps][Cout][Cin] bf16; bias fp32 [Cout] or NULL; residual bf16 with
 * the layout of out, or NULL; out [B,OHf,OWf,Cout] bf16, or fp32 when out_f32 (accumulate=1 adds to
 * the existing fp32 contents).  relu_mask (bf16, layout of out, or NULL): the result is zeroed where relu_mask <= 0 --
 * used by data-gradient launches to apply the ReLU backward of the tensor they differentiate (their own saved input)
 * in the epilogue instead of a separate pass.  The launch iterates oy<OH, ox<OW; source pixels outside
 * [0,IH)x[0,IW) contribute zero.  zero_page: >=256 bytes of device zeros.  Cin % 64 == 0,
 * Cout % 4 == 0, T <= 16.  groups/g: independent problems at the given element strides
 * (FOA rotation branches).  dy/dx/wt are HOST arrays. */
int loft_conv_tap_bf16(const void* src, const void* wgt, const float* bias, const void* residual, const void* relu_mask,
                       void* out,
                       const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf,
                       int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host,
                       const int* wt_host, int relu, int out_f32, int accumulate, int groups, int64_t src_gs,
                       int64_t wgt_gs, int64_t out_gs, int64_t bias_gs, void* stream);
/* Tail of a 64-plane ResNet bottleneck in one launch (mmdet/models/backbones/resnet.py:266-298: conv2 + bn2 + relu, conv3 + bn3,
 * `out += identity`, relu; with wd also `identity = self.downsample(x)`), for blocks nobody differentiates -- the frozen layer1
 * (`frozen_stages=1`, bonai_loft_foa_r50_fpn_basic.py:10) and inference:
 *   out[B,H,W,256] = relu( W3 . relu(conv3x3_pad1(t1) + b2) + b3 + shortcut ),
 * t1 [B,H,W,64] the first 1x1's output, w2 [9][64][64] / w3 [256][64] / wd [256][64] BN-folded forward packings, b2 [64], b3 [256]
 * fp32.  shortcut: wd == NULL -> the block input [B,H,W,256]; else the 64-channel block input x, and the shortcut conv
 * Wd . x + bd (bd [256]) is computed in the same launch and passes through the 16-bit type before the add, as between separate
 * launches: the output is bit-identical to the unfused block.  The 3x3's taps as (dy, dx, index into w2), each in -1..1. */
int loft_bneck_tail_bf16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* shortcut,
                         const void* wd, const float* bd, void* out, const void* zero_page, int B, int H, int W, const int* dy_host,
                         const int* dx_host, const int* wt_host, void* stream);
/* Two consecutive 1x1 convolutions of a ResNet stage in one launch (mmdet/models/backbones/resnet.py:266-298, Bottleneck.forward):
 * the END of block k (`out = self.conv3(out); out = self.norm3(out); out += identity; out = self.relu(out)`) and the BEGINNING of
 * block k+1 (`out = self.conv1(x); out = self.norm1(out); out = self.relu(out)`), so that the block output -- the widest map of
 * the stage -- is written to HBM once and NOT read back as conv1's operand (bneck_pair.hip).  All maps are [M][channels] rows
 * (NHWC with M = B*H*W, M % 128 == 0), P in {128, 256} planes, C % 128 == 0 (C = 4 P in ResNet):
 * weights in the K8 layout of loft_pack_k8_multi ([K/8][rows][8] of a [rows][K] packing):
 *   forward  (mask1 == NULL):  mid [M][C] = relu(w1 . a_in + bias1 + res),  out2 [M][P] = relu(w2 . mid + bias2)
 *                              a_in = t2_k [M][P], w1 = conv3_k forward packing [C][P], res = block k's shortcut [M][C],
 *                              w2 = conv1_{k+1} forward packing [P][C]; mid = out_k, out2 = t1_{k+1}; both biases required.
 *   backward (mask1 != NULL):  mid = (w1 . a_in + res) where mask1 > 0 else 0,  out2 = (w2 . mid) where mask2 > 0 else 0
 *                              a_in = d t1_{k+1} [M][P], w1 = conv1_{k+1} data-gradient packing [C][P], res = the gradient arriving
 *                              over block k+1's identity shortcut [M][C], mask1 = out_k, w2 = conv3_k data-gradient packing [P][C],
 *                              mask2 = t2_k [M][P]; mid = d out_k, out2 = d t2_k; no biases, both masks required.
 * Rounding points and fp32 operation order as in the separate launches (mid, out2 in the 16-bit type, fp32 accumulation, the second
 * product reads the rounded mid; bias and residual are added on the matrix pipe as exact products): the same bits as the separate
 * launches but for an element in ~5e5 (measured: none at P = 256, one in 524 288 at P = 128), one unit of the 16-bit type apart.
 * Anything else (shape, missing operand) returns hipErrorInvalidValue (1) and launches nothing. */
int loft_bneck_pair_bf16(const void* a_in, const void* w1, const float* bias1, const void* res, const void* mask1, void* mid,
                         const void* w2, const float* bias2, const void* mask2, void* out2, int64_t M, int P, int C, void* stream);
/* Re-arrange `n` 16-bit matrices [R][K] (K % 8 == 0) into the K8 layout [K/8][R][8] in one launch: desc = device array of n x 4
 * int64 {src, dst, R, K}; max_pieces = max over the matrices of R * K / 8.  (The weight operands of loft_bneck_pair_bf16: an MFMA
 * fragment is then 2 x 512 contiguous bytes instead of 64 pieces of 16 bytes.) */
int loft_pack_k8_multi(const int64_t* desc, int n, int64_t max_pieces, void* stream);
/* The same with a timing-ablation code (variant != 0: forward P = 256 instance with parts of the work removed -- RESULTS WRONG;
 * bneck_pair.hip ABL; tools/probes/pair_time.py).  variant 0 = loft_bneck_pair_bf16. */
int loft_bneck_pair_bf16_v(const void* a_in, const void* w1, const float* bias1, const void* res, const void* mask1, void* mid,
                           const void* w2, const float* bias2, const void* mask2, void* out2, int64_t M, int P, int C, int variant,
                           void* stream);
/* The same with the kernel chosen by the caller instead of the shape heuristics (tests pin every template the bench
 * dispatches; A/B timing).  variant = one LOFT_CONV_* kernel code, optionally OR-ed with LOFT_CONV_FLAG_*; LOFT_CONV_AUTO is
 * loft_conv_tap_bf16.  A kernel that cannot serve the shape returns hipErrorInvalidValue (1). */
#define LOFT_CONV_AUTO 0
#define LOFT_CONV_PIPE256 1      /* 256x256x64, two wave groups one barrier apart, counted vmcnt (Cout % 256 == 0) */
#define LOFT_CONV_T256_FAST 2    /* 256x256x64, lockstep double buffer, hoisted addressing (Cout % 256 == 0) */
#define LOFT_CONV_T256 3
#define LOFT_CONV_T128_SINGLE 4  /* 128x128x64, one LDS stage, 4 waves per SIMD (K-shallow launches; Cout % 128 == 0) */
#define LOFT_CONV_T128_FAST 5
#define LOFT_CONV_T128 6
#define LOFT_CONV_T128x64 7      /* any Cout % 4 == 0 */
#define LOFT_CONV_PATCH64 8      /* 64 -> 64 channels, stride 1, <= 3x3: halo patch kernel */
#define LOFT_CONV_STREAM256 9    /* 256x256x64, one software-pipelined stream per wave, one barrier per K-tile (Cout % 256 == 0) */
#define LOFT_CONV_STREAM128 10    /* the same stream kernel on 128-pixel x 256-cout tiles (launches with too few 256-pixel tiles) */
#define LOFT_CONV_STREAM64 11     /* ... on 64-pixel x 256-cout tiles (layer4's 32 x 32 maps) */
#define LOFT_CONV_STREAM64N 12    /* ... on 64-pixel x 128-cout tiles (256-channel convs on 32 x 32 maps: FPN P5, the RPN conv on it) */
#define LOFT_CONV_ROLES256 13     /* 256x256x64, role-split stream: waves 0-3 issue every activation copy, waves 4-7 every weight copy (three weight stages) */
#define LOFT_CONV_STREAM256N 14   /* the stream kernel on 256-pixel x 128-cout tiles with the THREE-stage ring also for Cout % 256 == 0 (1.5x the copy bytes per FLOP of the 256 x 256 tile, two K-tiles of look-ahead instead of one; A/B) */
#define LOFT_CONV_RING32 15       /* the stream kernel's 256 x 256 tile with 32-channel K-tiles on a FOUR-stage ring: pieces requested three tiles ahead, two per wave and sub-step, counted vmcnt (bit-identical to LOFT_CONV_STREAM256) */
#define LOFT_CONV_W4 16           /* the stream schedule's 256 x 256 x 64 tile with FOUR waves, one per SIMD, 128 x 128 each (conv_tap_w4_kernel; bit-identical to LOFT_CONV_STREAM256) */
#define LOFT_CONV_XFIRST 17       /* the two-stage 256 x 256 stream schedule with the operands' copy slots swapped: activation copies of K-tile t+2 requested right behind SYNC(t), weight copies of t+1 in ks0 (round 6; bit-identical to LOFT_CONV_STREAM256) */
#define LOFT_CONV_LEAN 18         /* the two-stage 256 x 256 stream schedule with fewer instructions per K-tile: compile-time chunk-major sequencer, one 64-bit scalar offset per K-tile, select-free activation copies on all-valid tiles (round 6; chunk-major launches only; bit-identical to LOFT_CONV_STREAM256) */
#define LOFT_CONV_LEANX 19        /* LOFT_CONV_LEAN + LOFT_CONV_XFIRST */
/* The form of the two-stage 256 x 256 stream schedule the dispatcher itself launches: 0 round 2's, 1 LOFT_CONV_XFIRST, 2 LOFT_CONV_LEAN,
 * 3 LOFT_CONV_LEANX (tap-major launches keep 0 / 1).  loft_conv_stream_form(form) sets it process-wide and returns the previous value
 * (form < 0: query only) -- for same-box A/B runs of a whole step; results are bit-identical under every form.  form + 4: the
 * operand-plane launches of the 256 x 256 tile keep the direct fp32 epilogue instead of the LDS-staged one (round 6; bit-identical too). */
#define LOFT_STREAM_FORM_DEFAULT 2
int loft_conv_stream_form(int form);
#define LOFT_CONV_FLAG_NO_PIXMAJOR 0x100
#define LOFT_CONV_FLAG_NO_NFAST 0x200
#define LOFT_CONV_FLAG_NO_STAGED_OUT 0x400
#define LOFT_CONV_FLAG_TAP_MAJOR 0x800   /* pipelined kernels: K order (tap, channel chunk) -- the lock-step kernels' order -- instead of (chunk, tap) */
#define LOFT_CONV_FLAG_KROT 0x20000   /* pipelined kernels, chunk-major order: rotate the channel-chunk order per workgroup (see ConvArgs::krot) */
#define LOFT_CONV_FLAG_NO_ROI_BLOCKS 0x10000   /* pipelined kernels on RoI maps: position-major rows over all RoIs instead of blocks of ~256 RoIs */
int loft_conv_tap_bf16_v(const void* src, const void* wgt, const float* bias, const void* residual, const void* relu_mask,
                         void* out,
                         const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf,
                         int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host,
                         const int* wt_host, int relu, int out_f32, int accumulate, int groups, int64_t src_gs,
                         int64_t wgt_gs, int64_t out_gs, int64_t bias_gs, int variant, void* stream);

/* loft_conv_tap_bf16_head: loft_conv_tap_bf16 (one group, bf16 output, the library's own kernel choice) with a NARROW 1x1 HEAD on its
 * output computed in the epilogue, from the bf16 tile before it leaves LDS: head_out[pixel][n] = head_b[n] + sum_c bf16(out[pixel][c]) *
 * head_w[n][c] for n < head_c4 (a multiple of 4, <= 32).  head_w: activation type [head_c4][256], head_b
 * fp32 [head_c4], head_out fp32 [pixels of the FULL output map (B*OHf*OWf)][head_c4].  Replaces a second launch that reads the wide map
 * back for a handful of outputs: the RPN's objectness + delta convs behind its 3x3 conv (rpn_head.py:38-44), the mask logits behind the
 * 2x2 deconvolution (fcn_mask_head.py:121-126).  Served by the 256-cout stream tiles only (Cout == 256): any other launch returns
 * hipErrorInvalidValue WITHOUT launching anything -- the caller then runs the head as a launch of its own. */
int loft_conv_tap_bf16_head(const void* src, const void* wgt, const float* bias, const void* residual, const void* relu_mask, void* out,
                            const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf, int OWf, int os,
                            int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host, const int* wt_host, int relu,
                            const void* head_w, const float* head_b, float* head_out, int head_c4, void* stream);

/* loft_deconv2x2_bf16: ConvTranspose2d(kernel 2, stride 2) + bias (+ ReLU) in ONE launch (fcn_mask_head.py:121-124): out[b, 2y + py,
 * 2x + px, :] = act(W[2 py + px] . src[b, y, x, :] + bias); src [B,H,W,Cin], wgt [4][256][Cin] (tap p = 2 py + px, forward packing),
 * out [B,2H,2W,256], activation type.  The four taps are four channel tiles of the same pixel tile: src is read from HBM once, not
 * once per output parity.  head_w != NULL: the narrow 1x1 head of loft_conv_tap_bf16_head on the output (head_out fp32
 * [B*2H*2W][head_c4]).  Served for Cout == 256 and B*H*W >= 192 * 256 pixels; otherwise hipErrorInvalidValue, nothing launched. */
int loft_deconv2x2_bf16(const void* src, const void* wgt, const float* bias, void* out, const void* zero_page, int B, int H, int W,
                        int Cin, int Cout, int relu, const void* head_w, const float* head_b, float* head_out, int head_c4,
                        void* stream);
/* fp32 parity mode, backward (parity_f32.hip): weight gradient of the same tap contract on v_mfma_f32_32x32x2_f32 (dw is
 * accumulated into: the caller zeroes it), and the fp32 forms of the glue adjoints.  Checker path (1e-3 vs the reference's fp32
 * autograd), not a performance path. */
/* Contraction of the fp32 parity mode (loft_conv_tap_f32_v / loft_conv_wgrad_f32_v; the plain names take the default), fp32
 * operands and fp32 accumulation in every case:
 * LOFT_F32_SPLIT6 (default since round 4): every operand element = hi + mid + lo, three bf16 (24 mantissa bits), a product = the
 *   six v_mfma_f32_32x32x16_bf16 terms down to 2^-16 -- fp32-grade results at ~2x the fp32 MFMA kernels' speed;
 * LOFT_F32_SPLIT3: two bf16 per element (16 mantissa bits), three terms -- ~2e-6 of an output's scale per layer, ~2.5x; meets the
 *   1e-3 clause on losses, features and inference results, NOT on parameter gradients of a random-weight network (norms 2e-3,
 *   single entries up to 5e-2);
 * LOFT_F32_EXACT: v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain (rounds 1-3). */
#define LOFT_F32_SPLIT6 0
#define LOFT_F32_EXACT 1
#define LOFT_F32_SPLIT3 2
int loft_conv_wgrad_f32_v(const float* g, const float* x, float* dw, int B, int GH, int GW, int Cout, int XH, int XW, int Cin, int OH,
                          int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host, const int* dy_host,
                          const int* dx_host, const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                          int variant, void* stream);
int loft_conv_wgrad_f32(const float* g, const float* x, float* dw, int B, int GH, int GW, int Cout, int XH, int XW, int Cin, int OH,
                        int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host, const int* dy_host,
                        const int* dx_host, const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                        void* stream);
int loft_relu_bwd_f32(const float* g, const float* y, float* out, int64_t n, float* amax_out /* optional, PRE-ZEROED: max |out| */, void* stream);
int loft_downsum2x_add_f32(float* coarse, const float* fine, int B, int Hc, int Wc, int Hf, int Wf, int C, void* stream);
int loft_subsample2_add_f32(float* big, const float* small, int B, int Hs, int Ws, int Hb, int Wb, int C, void* stream);
/* loft_conv_tap_f32: the fp32 parity mode of the same contract (all operands and the output fp32, contraction on
 * v_mfma_f32_32x32x2_f32 = exact fp32 products and sums).  Forward / data-gradient only; it exists so inference results can
 * be checked against the reference's fp32 outputs at the north-star tolerance (1e-3), not for speed.  Cin % 32 == 0. */
int loft_conv_tap_f32(const float* src, const float* wgt, const float* bias, const float* residual, const float* relu_mask,
                      float* out, const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf,
                      int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host,
                      const int* wt_host, int relu, int accumulate, int groups, int64_t src_gs, int64_t wgt_gs,
                      int64_t out_gs, int64_t bias_gs, void* stream);
int loft_conv_tap_f32_v(const float* src, const float* wgt, const float* bias, const float* residual, const float* relu_mask,
                      float* out, const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf,
                      int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host,
                      const int* wt_host, int relu, int accumulate, int groups, int64_t src_gs, int64_t wgt_gs,
                      int64_t out_gs, int64_t bias_gs, int variant, void* stream);
/* The fp32 parity mode on OPERAND PLANES (round 5) -- the same contracts as loft_conv_tap_f32 / loft_conv_wgrad_f32, served by the
 * software-pipelined 16-bit kernels (conv_pipe.hip, conv_wgrad_pipe.hip) instead of kernels of their own.  Reference lines: every
 * nn.Conv2d / nn.Linear of the LOFT path evaluated in fp32 (detectors/base.py:159-173 before its fp16 cast; the reference's CPU path).
 *   loft_planes_per_tensor()      NP of this build: 2 (binary16 build) or 3 (bfloat16 build)
 *   loft_absmax_f32(x, n, amax)   amax[0] = max(amax[0], max |x|) as a device scalar (n % 4 == 0; the caller hands a zeroed word); inf when x holds a NaN
 *   loft_split_planes_f32(x, n, planes, amax)   planes [NP][n] of the build's 16-bit type with
 *                                 x * scale = plane_0 + plane_1 (+ plane_2), plane_k = RNE16 of the remainder; scale = the power of two
 *                                 that puts amax into [2^14, 2^15) (amax NULL, 0, inf: 1).  binary16 build: 22 significant bits per
 *                                 element, the scale is REQUIRED for tensors outside binary16's exponent range; bfloat16 build: 24 bits,
 *                                 pass NULL.  n % 8 == 0.
 *   loft_conv_tap_planes(...)     loft_conv_tap_bf16's contract with src / wgt = such planes (plane p at element offset p * x_ps /
 *                                 p * w_ps), `nterms` products (activation plane xpl[i], weight plane wpl[i]) accumulated in fp32 in ONE
 *                                 K loop; bias / residual / relu_mask / out fp32; the result is scaled by 1 / (scale_x * scale_w) taken from
 *                                 amax_x / amax_w (both or neither).  The terms the mode uses: binary16 (1,0) (0,1) (0,0) -- error 2^-22 of
 *                                 |x||w| per product; bfloat16 (0,2) (2,0) (1,1) (0,1) (1,0) (0,0) -- 2^-24.  hipErrorInvalidValue for
 *                                 shapes the stream kernel does not serve (Cout % 128, Cin % 64, nterms * T > 64): take loft_conv_tap_f32.
 *   loft_conv_wgrad_planes(...)   loft_conv_wgrad_bf16's contract on G / X planes: every term adds into dw (caller zeroes) through the
 *                                 split-K atomics, scaled by 1 / (scale_g * scale_x).  Cout % 128 == 0, Cin % 128 == 0.  db (may be NULL):
 *                                 the fused bias gradient of loft_conv_wgrad_bf16 -- the column sums of every G plane, each by the one
 *                                 term that pairs it with X plane 0 (hipErrorInvalidValue if the term list has no such pairing), scaled by
 *                                 1 / scale_g (round 6: replaces a loft_colsum_f32 pass over the fp32 gradient map). */
int loft_planes_per_tensor(void);
int loft_absmax_f32(const float* x, int64_t n, float* amax_out /* PRE-ZEROED by the caller */, void* stream);
int loft_split_planes_f32(const float* x, int64_t n, void* planes, const float* amax, void* stream);
/* absmax + split of one tensor.  slot: two PRE-ZEROED 32-bit words (the caller's pool: one memset per few thousand tensors);
 * slot[0] holds the absmax afterwards -- the amax_* argument of the contraction entry points.  Up to 4 Mi elements ONE launch (a
 * co-resident grid of <= 256 workgroups meets at a counter between the two phases), above that two.  n % 8 == 0. */
int loft_absmax_split_planes_f32(const float* x, int64_t n, void* planes, float* slot, void* stream);
/* out[g][c] += sum_r x[g][r][c]  (x fp32 [groups][rows][C], C % 4 == 0; out fp32 [groups][C], accumulated into): the bias gradient
 * of the fp32 parity mode's convolutions / linear layers (sum over pixels of the NHWC output gradient). */
int loft_colsum_f32(const float* x, int64_t rows, int C, int groups, float* out, void* stream);
int loft_conv_tap_planes(const void* src, const void* wgt, const float* bias, const float* residual, const float* relu_mask,
                         float* out, const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW, int OHf,
                         int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host, const int* dx_host,
                         const int* wt_host, int relu, int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs,
                         int nterms, const int* xpl_host, const int* wpl_host, int64_t x_ps, int64_t w_ps, const float* amax_x,
                         const float* amax_w, float* amax_out /* optional, PRE-ZEROED: max |out| of the stored elements */, void* stream);
int loft_conv_wgrad_planes(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH, int GW, int Cout, int XH,
                           int XW, int Cin, int OH, int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host,
                           const int* dy_host, const int* dx_host, const int* wt_host, int groups, int64_t g_gs, int64_t x_gs,
                           int64_t dw_gs, int nterms, const int* gpl_host, const int* xpl_host, int64_t g_ps, int64_t x_ps,
                           const float* amax_g, const float* amax_x, float* db, int db_tap, void* stream);
/* loft_conv_wgrad_bf16: weight gradient of the same family (autograd of the call sites above):
 *   dw[g][wt[t]][n][c] += sum_{b,oy,ox} G[g][b, oy*gos+goy[t], ox*gos+gox[t], n] * X[g][b, oy*ss+dy[t], ox*ss+dx[t], c]
 * G [B,GH,GW,Cout] bf16 (output gradient), X [B,XH,XW,Cin] bf16 (saved input), dw fp32
 * [taps][Cout][Cin] accumulated with atomics (caller zeroes).  Cin % 8 == 0, Cout % 8 == 0 (multiples of 128 run the wide tiles;
 * anything else the 64-channel narrow kernel, e.g. HRNet's 32/64-channel branches).
 * splits <= 0 lets the library choose the split-K factor.  db (may be NULL): fp32 [groups][Cout] bias
 * gradient sum_pixels G, accumulated in the same pass from tap db_tap (a tap whose X gather never
 * leaves the image, e.g. the centre tap; -2 = from every tap, for the transposed-conv case where the
 * taps partition the G pixels); caller zeroes. */
int loft_conv_wgrad_bf16(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH, int GW,
                         int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                         const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                         const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs, int splits,
                         float* db, int db_tap, void* stream);
/* The same with the kernel chosen by the caller (tests / A-B timing): LOFT_WGRAD_AUTO = loft_conv_wgrad_bf16's heuristics. */
#define LOFT_WGRAD_AUTO 0
#define LOFT_WGRAD_STREAM256 1   /* 256x256 tile, software-pipelined stream (Cout, Cin multiples of 256) */
#define LOFT_WGRAD_T256 2        /* 256x256 tile, lockstep double buffer */
#define LOFT_WGRAD_T128 3        /* 128x128 tile (Cout, Cin multiples of 128), lock-step double buffer */
#define LOFT_WGRAD_RING128 4     /* 128x128 tile, four-stage ring of 32-pixel K-steps (the AUTO choice where the 256 forms do not apply) */
int loft_conv_wgrad_bf16_v(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH, int GW,
                         int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                         const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                         const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs, int splits,
                         float* db, int db_tap, int variant, void* stream);
/* Split-K without atomics.  An fp32 atomic costs the L2 a channel-cycle per 4 bytes (0.29 T atomics/s over the chip), a store
 * 1/16 of that, so the weight gradient can also be left as per-split partial sums: loft_conv_wgrad_slots -> the number of slots S
 * such a launch of this shape writes (0: no such form -- narrow channels, repeated / missing weight taps, a tap without a valid
 * row; < 0: -hipError_t); loft_conv_wgrad_bf16_slots writes dw_slots = fp32 [groups][S][T][Cout][Cin] completely (nothing needs to
 * be zeroed; db is still accumulated with atomics into a zeroed buffer); loft_fold_unpack_bwd_multi sums the S slots while it
 * reads (descriptor word 9: eps bits | S << 32). */
int loft_conv_wgrad_slots(int B, int GH, int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                          const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host, const int* wt_host,
                          int groups, int splits, int variant);
int loft_conv_wgrad_bf16_slots(const void* g, const void* x, float* dw_slots, const void* zero_page, int B, int GH, int GW,
                               int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T, const int* goy_host,
                               const int* gox_host, const int* dy_host, const int* dx_host, const int* wt_host, int groups,
                               int64_t g_gs, int64_t x_gs, int nslots, int splits, float* db, int db_tap, int variant,
                               void* stream);

/* loft_conv_wgrad_patch_bf16: the same weight (+ bias) gradient for stride-1, same-size convs with <= 64 input and output
 * channels and taps within +-1 pixel (HRNet's high-resolution 3x3 branches, hrnet.py:12-60 via resnet.py:13-92 BasicBlock): every
 * workgroup walks 8x8-pixel patches and feeds ALL taps from one staged patch of g and its 10x10 halo of x.  g [groups*B,H,W,Cout],
 * x [groups*B,H,W,Cin] bf16 NHWC; dw [groups][T][Cout][Cin] and db [groups][Cout] (may be NULL) fp32 are accumulated into;
 * workspace: loft_conv_wgrad_patch_workspace_bytes(...) bytes for the per-workgroup partial sums. */
int64_t loft_conv_wgrad_patch_workspace_bytes(int B, int H, int W, int Cout, int Cin, int T, int groups);
int loft_conv_wgrad_patch_bf16(const void* g, const void* x, float* dw, const void* zero_page, int B, int H, int W,
                               int Cout, int Cin, int T, const int* dy_host, const int* dx_host, const int* wt_host,
                               int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs, float* db, void* workspace, void* stream);


/* ---- HBM-bound glue (bf16 NHWC unless noted; n = element counts, multiples of 8) ------------
 * relu_bwd: out = g * (y > 0)  -- autograd of the ReLUs fused into the conv epilogues
 *           (mmdet/models/backbones/resnet.py:266-298 `self.relu`, mmcv ConvModule activations).
 * colsum:   out[c] += sum_m x[m][c] (fp32 atomics; bias / frozen-BN beta gradients).
 * upsample2x_add / downsum2x_add: FPN top-down `laterals[i-1] += F.interpolate(laterals[i],
 *           mode='nearest')` (mmdet/models/necks/fpn.py:176-181) and its adjoint.
 * subsample2: P6 = F.max_pool2d(P5, 1, stride=2) (fpn.py:189-191); adjoint=1 scatter-adds back.
 * maxpool3x3s2: nn.MaxPool2d(3, 2, 1) of the stem (resnet.py:631), forward only (frozen stage).
 * stem7x7_bn_relu: conv1 7x7/2 + frozen bn1 + relu (resnet.py:628-630): img fp32 NCHW [B,3,H,W],
 *           w fp32 [64,3,7,7], scale/shift fp32 [64] (folded BN) -> bf16 NHWC [B,H/2,W/2,64]. */
int loft_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t n, void* stream);
int loft_colsum_bf16(const void* x, int64_t M, int C, float* out, void* stream);
int loft_upsample2x_add_bf16(void* fine, const void* coarse, int B, int H, int W, int C, void* stream);
int loft_downsum2x_add_bf16(void* coarse, const void* fine, int B, int Hc, int Wc, int C, void* stream);
/* out-of-place form of the same sum (out = coarse + 2x2 block sums of fine, same order of additions): the FPN backward keeps the
 * incoming coarse gradient untouched without cloning it first. */
int loft_downsum2x_sum_bf16(void* out, const void* coarse, const void* fine, int B, int Hc, int Wc, int C, void* stream);
int loft_subsample2_bf16(const void* src, void* dst, int B, int Ho, int Wo, int Hi, int Wi, int C, int adjoint,
                         void* stream);
int loft_maxpool3x3s2_bf16(const void* src, void* dst, int B, int Hi, int Wi, int C, void* stream);
int loft_stem7x7_bn_relu(const float* img, const float* w, const float* scale, const float* shift, void* out, int B,
                         int H, int W, int out_f32, void* stream);
/* fp32 parity-mode forms of the three forward glue kernels above (same reference lines). */
int loft_upsample2x_add_f32(float* fine, const float* coarse, int B, int H, int W, int C, void* stream);
int loft_subsample2_f32(const float* src, float* dst, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream);
int loft_maxpool3x3s2_f32(const float* src, float* dst, int B, int Hi, int Wi, int C, void* stream);
/* MFMA form of the stem (same reference lines): wgt_packed bf16 [64][192], row n = the 147 weights of output
 * channel n in (c, r, s) order times the folded BN scale, zero padded; bias fp32 [64] = folded BN shift. */
int loft_stem7x7_mfma(const float* img, const void* wgt_packed, const float* bias, void* out, int B, int H, int W,
                      void* stream);
int loft_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int loft_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* Optimizer step of the reference run (mmcv OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)) +
 * torch.optim.SGD(lr, momentum=0.9, weight_decay=1e-4); configs/_base_/schedules/schedule_2x_bonai.py:2-3)
 * on one flat fp32 parameter arena: sumsq accumulates sum(g^2) into *out (caller zeroes);
 * sgd applies g*grad_scale, the clip factor from sqrt(*gnorm_sq)*grad_scale, weight decay, momentum. */
int loft_sumsq_f32(const float* g, int64_t n, float* out, void* stream);
int loft_sgd_momentum_f32(float* p, const float* g, float* m, int64_t n, const float* gnorm_sq, float max_norm, float lr,
                          float momentum, float weight_decay, float grad_scale, void* stream);

/* ---- box / target arithmetic (fp32 + integer) -------------------------------------------------
 * iou_assign: MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:60-212 with
 *   gt_max_assign_all=True, no ignore regions) for B images at once: boxes [B,Nmax,4] (nbox[b] valid),
 *   gts [B,Kmax,4] (ngt[b] valid) -> gt_inds int64 [B,Nmax] (0 neg, -1 ignore/padding, i+1 pos),
 *   max_ov fp32 [B,Nmax].  argmax_ws int32 [B,Nmax], gt_max_ws uint32 [B,Kmax] are workspaces. */
int loft_iou_assign(const float* boxes, const int* nbox, int Nmax, const float* gts, const int* ngt, int Kmax, int B,
                    float pos_thr, float neg_thr, float min_pos, int low_quality, float* max_ov, int32_t* argmax_ws,
                    uint32_t* gt_max_ws, int64_t* gt_inds, void* stream);
/* DeltaXYWHBBoxCoder.decode / .encode (mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:78-197), [n,4] each;
 * max_w <= 0 disables the clamp. */
int loft_delta2bbox(const float* rois, const float* deltas, int64_t n, const float* means_host, const float* stds_host,
                    float wh_ratio_clip, float max_h, float max_w, float* out, void* stream);
int loft_bbox2delta(const float* proposals, const float* gt, int64_t n, const float* means_host, const float* stds_host,
                    float* out, void* stream);
/* RPN candidate generation for one pyramid level of all B images (rpn_head.py:116-150):
 * head fp32 [B,H,W,Cp]: channel a<A objectness logit, channel A+4a+j delta j.  rpn_scores writes
 * sigmoid scores into keys[b*img_stride + lvl_off + pos*A + a]; after loft_segmented_sort_desc over
 * the (image, level) segments, rpn_decode turns ranks r<topk into boxes (anchor = base_anchors[a] +
 * (x*stride, y*stride), decoded and clamped) at out_boxes[b*cand_stride + cand_off + r]. */
int loft_rpn_scores(const float* head, int B, int H, int W, int Cp, int A, int64_t img_stride, int64_t lvl_off,
                    float* keys, void* stream);
int loft_rpn_decode(const float* head, const int32_t* sorted_idx, int B, int H, int W, int Cp, int A, int64_t img_stride,
                    int64_t lvl_off, int topk, const float* base_anchors, int stride, const float* means_host,
                    const float* stds_host, float wh_ratio_clip, float max_h, float max_w, int64_t cand_stride,
                    int64_t cand_off, float* out_boxes, void* stream);
/* The same two steps for every pyramid level in ONE launch each, plus what rpn_head.py:133-168 does around them: the decode also
 * writes the candidates' scores in the candidate layout (out_scores [B][cand_stride], may be NULL) and accumulates the per-image
 * maximum coordinate (img_max [B], may be NULL; reset to -inf by loft_rpn_scores_levels) that batched_nms shifts the levels by.
 * Host arrays of `levels` (<= 8) entries; heads fp32 NHWC [B,H_l,W_l,Cp]. */
int loft_rpn_scores_levels(const void* const* heads_host, const int* H_host, const int* W_host, const int64_t* lvl_off_host,
                           int levels, int B, int Cp, int A, int64_t img_stride, float* keys, float* img_max, void* stream);
int loft_rpn_decode_levels(const void* const* heads_host, const void* const* base_anchors_host, const int* H_host, const int* W_host,
                           const int* topk_host, const int* stride_host, const int64_t* lvl_off_host, const int64_t* cand_off_host,
                           int levels, const int32_t* sorted_idx, const float* sorted_keys, int B, int Cp, int A, int64_t img_stride,
                           const float* means_host, const float* stds_host, float wh_ratio_clip, float max_h, float max_w,
                           int64_t cand_stride, float* out_boxes, float* out_scores, float* img_max, void* stream);
/* loft_rpn_finalize: rpn_head.py:169-171 (`dets[:cfg.nms_post]` per image) from the score-sorted survivor list: top_scores / top_idx
 * = loft_segmented_topk_desc over segments of seg_stride candidates (suppressed ones carry -1), top_idx into cand_boxes
 * [B * seg_stride][4] -> props [B][post][5] (box, score; rows past the survivors zero), counts int64 [B]. */
int loft_rpn_finalize(const float* top_scores, const int32_t* top_idx, const float* cand_boxes, int B, int64_t seg_stride, int post,
                      float* props, int64_t* counts, void* stream);
/* FOA: 4-rotation offset targets (offset_head_expand_feature.py:271-344 + delta_xy_offset_coder.py:46-65)
 * -> out [4n,2] branch-major; and inference fusion + decode (:346-448) pred [4n,2] -> out [n,2]. */
int loft_foa_targets(const float* pos_boxes, const float* pos_gt_offsets, int64_t n, float std_x, float std_y, float* out,
                     void* stream);
int loft_foa_fuse_decode(const float* pred, const float* boxes, int64_t n, float std_x, float std_y, float max_h,
                         float max_w, float* out, void* stream);
/* Plain LOFT OffsetHead without FOA (attribute_heads/offset_head.py:118-188 get_targets, :190-243 get_offsets;
 * DeltaXYOffsetCoder delta_xy_offset_coder.py:46-88).  targets: out [n,reg_num]; reg_num 2 = encoded (dx,dy), reg_num 3 =
 * (length, cos(angle), sin(angle)) of the encoded pair (:176-183).  decode: pred [n,reg_num] -> out [n,2] pixels, clamped to
 * +-(max_w,max_h); polar != 0 = offset_coordinate 'polar' (length*(cos,sin) of the decoded pair, :232-236). */
int loft_offset_targets(const float* pos_boxes, const float* pos_gt_offsets, int64_t n, float mean_x, float mean_y,
                        float std_x, float std_y, int reg_num, float* out, void* stream);
int loft_offset_decode(const float* pred, const float* boxes, int64_t n, float mean_x, float mean_y, float std_x, float std_y,
                       float max_h, float max_w, int reg_num, int polar, float* out, void* stream);
/* Polygon -> instance bitmaps on the device: LoadAnnotations._poly2mask (mmdet/datasets/pipelines/loading.py:301-326,
 * pycocotools frPyObjects + merge + decode; the rleFrPoly algorithm restated, [pycocotools not in tree]).  xy: fp64 vertices
 * [total_vertices, 2] of all polygons; poly_offsets int64 [P+1] (vertex ranges); inst_poly_offsets int64 [K+1] (polygon ranges of
 * the K instances: an instance's polygons are OR-merged); out uint8 [K, H, W] (every byte written).  All DEVICE pointers.
 * Limits: H*W <= 1024*1024 + slack (the toggle bitmap lives in LDS), max_vertices (largest polygon) <= 1023. */
int loft_poly2mask(const double* xy, const int64_t* poly_offsets, const int64_t* inst_poly_offsets, int num_inst, int H, int W,
                   int max_vertices, uint8_t* out, void* stream);
/* Mask targets on device (mmdet/core/mask/mask_target.py:33-62 -> structures.py:261-291):
 * masks u8 [K,H,W]; RoI i crops mask gt_idx[i] with box boxes[i] (clipped to the image) to SxS,
 * RoIAlign(avg, aligned, adaptive grid) >= 0.5 -> out fp32 {0,1} [n,S,S].  mask_addr (optional, device int64 [#instances]): the address of every instance mask -- gt_idx then
 * indexes it and per-image mask tensors need no concatenation (masks may be NULL). */
int loft_mask_target(const uint8_t* masks, int H, int W, const float* boxes, const int64_t* gt_idx, int64_t n, int S,
                     float* out, const int64_t* mask_addr, void* stream);

/* ---- inference post-processing --------------------------------------------------------------------
 * loft_soft_nms: mmcv.ops.soft_nms (CPU-only in mmcv 1.0.5), reached from multiclass_nms
 * (mmdet/core/post_processing/bbox_nms.py:63, test_cfg.rcnn.nms = dict(type='soft_nms', iou_threshold=0.5)).
 * boxes [n,4], scores [n] -> dets [n,5] and inds int64 [n] in selection order, *n_out kept.  method 0 naive,
 * 1 linear, 2 gaussian.  Same in-place max-selection order (ties: first position) as the reference op.
 * loft_mask_paste: FCNMaskHead.get_seg_masks / _do_paste_mask (fcn_mask_head.py:151-308): logits fp32 [N,S,S]
 * (class already selected), boxes [N,4] in output-image pixels -> out uint8 {0,1} [N,img_h,img_w]. */
int64_t loft_soft_nms_workspace_bytes(int64_t n);
int loft_soft_nms(const float* boxes, const float* scores, int64_t n, float iou_thr, float sigma, float min_score, int method,
                  void* workspace, float* dets, int64_t* inds, int* n_out, void* stream);
int loft_mask_paste(const float* logits, const float* boxes, int N, int S, int img_h, int img_w, float thr, uint8_t* out,
                    void* stream);
/* loft_mask_translate: footprint bitmaps from roof bitmaps and predicted offsets -- what the reference's evaluation does on polygons
 * through the external bstool package (tools/bonai/bonai_evaluation.py:64-91 BSPklParser(..., offset_model='footprint2roof'):
 * footprint = roof translated by -offset; offsets are the third element of the result tuples of mmdet/apis/test.py:53-72).
 * masks uint8 [N,H,W], offsets fp32 [N,2] = (dx, dy) in pixels -> out[n, y, x] = masks[n, y + round(dy), x + round(dx)] (0 outside). */
int loft_mask_translate(const uint8_t* masks, const float* offsets, int N, int H, int W, uint8_t* out, void* stream);

/* ---- sparse backward of the RPN head -------------------------------------------------------------
 * The RPN losses (anchor_head.py:429-497, rpn_head.py:56-80) read the head outputs only at the sampled anchors (<= 256 per
 * image of 261 888), so its backward runs on the selected pixels: level_ptrs/H/W (HOST arrays, n_levels <= 8) describe the
 * bf16 NHWC pyramid maps [B,H_l,W_l,C]; rows int32 [nsel][4] = (b, level, y, x) on the device, level < 0 = inactive.
 * gather: out bf16 [nsel][K*K][C] = the K x K neighbourhood rows (zeros outside the map): the A operand of the dense
 * weight-gradient GEMM.  scatter_add: map[level][b, y+dy, x+dx, :] += src[nsel][K*K][C] with packed bf16 atomics: the
 * data gradient.  K odd. */
int loft_rpn_gather_rows(void* const* level_ptrs, const int* H, const int* W, int n_levels, const int* rows, int nsel, int C,
                         int K, void* out, void* stream);
int loft_rpn_scatter_add_rows(void* const* level_ptrs, const int* H, const int* W, int n_levels, const int* rows, int nsel,
                              int C, int K, const void* src, void* stream);

/* loft_rpn_sparse_prep: the operands of the RPN head's sparse backward (rpn_head.py:38-54 under anchor_head.py:429-497) in one launch.
 * g fp32 [nsel,5] = gradient of (objectness logit, 4 deltas) of every sampled anchor, slot int64 [nsel] = its anchor index a < A;
 * w_cls fp32 [A,C], w_reg fp32 [4A,C], w_conv fp32 [C,C,3,3].  Outputs in the activation type of the build:
 * g_rows [nsel,P] (column a = g[.,0], columns A+4a+j = g[.,1+j], zeros elsewhere; P >= 5A), w_headT [C,P] (the fused 1x1 head's
 * weights transposed, zero-padded to P), wd [9C,C] with wd[t C + ci][co] = w_conv[co][ci][t]. */
int loft_rpn_sparse_prep(const float* g, const int64_t* slot, int nsel, int A, int P, int C, const float* w_cls, const float* w_reg,
                         const float* w_conv, void* g_rows, void* w_headT, void* wd, void* stream);

/* ---- HRNet-W32 / HRFPN resampling and fusion (BASELINE config 5) ----------------------------------
 * All tensors NHWC, dtype LOFT_F32 | LOFT_BF16, C % 8 == 0, scale factors are 1 << shift.
 * loft_fuse_sum_relu: out = relu?(sum_j nearest_up(terms[j], 1 << shifts[j])) -- the fuse step of HRModule.forward
 *   (mmdet/models/backbones/hrnet.py:177-195; nn.Upsample(mode='nearest') of :141-143); terms[j] is [B, H>>s_j, W>>s_j, C];
 *   terms / shifts are HOST arrays, n_terms <= 4.
 * loft_blocksum_masked: out[B,Hc,Wc,C] = sum over each (1<<shift)^2 block of g * (y > 0) (y may be NULL): its backward.
 * loft_bilinear_up_slot: F.interpolate(x, scale_factor=1<<shift, mode='bilinear') (align_corners=False) written into
 *   channels [coff, coff+C) of a [B, h<<shift, w<<shift, Ctot] tensor -- HRFPN.forward's upsample + torch.cat
 *   (mmdet/models/necks/hrfpn.py:79-85); backward != 0: src is the gradient of the slotted tensor, dst the gradient of x.
 * loft_avgpool: F.avg_pool2d(x, 1<<shift, 1<<shift) (hrfpn.py:90-92); backward != 0: src = gradient of the pooled map,
 *   dst = gradient of x (accumulate != 0 adds to dst).
 * loft_stem3x3s2_bn_relu: HRNet stem conv1 3x3/2 (3 -> 64) + frozen-stat BN (scale/shift) + ReLU from the fp32 NCHW image
 *   (hrnet.py:273-281, 481-483).  loft_stem3x3s2_wgrad: dwp fp32 [9][64][3] += gradient of the BN-folded weight, db fp32
 *   [64] += gradient of the folded bias, from g and the saved output y (ReLU mask applied here); caller zeroes both and
 *   feeds them to loft_fold_unpack_bwd. */
int loft_fuse_sum_relu(const void* const* terms, const int* shifts, int n_terms, void* out, int dtype, int B, int H, int W,
                       int C, int relu, void* stream);
int loft_blocksum_masked(const void* g, const void* y, void* out, int dtype, int B, int Hc, int Wc, int C, int shift,
                         void* stream);
int loft_bilinear_up_slot(const void* src, void* dst, int dtype, int B, int h, int w, int C, int shift, int Ctot, int coff,
                          int backward, void* stream);
int loft_avgpool(const void* src, void* dst, int dtype, int B, int Ho, int Wo, int C, int shift, int backward, int accumulate,
                 void* stream);
int loft_stem3x3s2_bn_relu(const float* img, const float* w, const float* scale, const float* shift, void* out, int dtype,
                           int B, int H, int W, void* stream);
int loft_stem3x3s2_wgrad(const float* img, const void* g, const void* y, float* dwp, float* db, int dtype, int B, int H, int W,
                         void* stream);

/* ---- modulated deformable convolution (DCNv2) sampling ------------------------------------------
 * Replaces mmcv.ops.ModulatedDeformConv2dPack / modulated_deform_conv2d [mmcv==1.0.5, not in tree] at the call sites
 * mmdet/models/backbones/resnet.py:171-194 (Bottleneck.conv2 when dcn=dict(type='DCNv2')) and
 * mmdet/models/necks/fpn.py:116-132 (conv_cfg=dict(type='DCNv2')).
 * x: NHWC [B,IH,IW,C] (dtype: LOFT_F32 | LOFT_BF16); offmask: fp32 NHWC [B,OH,OW,offmask_stride], the RAW conv_offset output
 * (channels [0, 2*DG*K) = offsets, channel g*2K+2k = dy and +1 = dx of tap k in group g; [2*DG*K, 3*DG*K) = mask logits,
 * sigmoid applied here).  col: [B*OH*OW][K][C] (same dtype as x) = mask * zero-padded bilinear sample, i.e. the A operand of
 * the 1x1 contraction with the weight packed as [Cout][K*C] that loft_conv_tap_* then runs.
 * bwd: dcol (layout of col) -> dx fp32 [B,IH,IW,C] (ACCUMULATED with atomics: caller zeroes), doffmask fp32 (layout of offmask;
 * channels >= 3*DG*K untouched).  K = kh*kw <= 9, DG <= 4, C % (8*DG) == 0 and C/(8*DG) a power of two.
 * workspace: loft_mdcn_bwd_workspace_bytes() bytes; 0 / NULL = the generic kernel that scatters with global fp32 atomics
 * (deform_groups > 1, C % 64 != 0).  With a workspace the tiled, atomic-free path runs (LDS gradient windows per output
 * tile, stored to the workspace and summed per input pixel by a second pass; doffmask is fully overwritten). */
int loft_mdcn_sample_fwd(const void* x, const float* offmask, void* col, int dtype, int B, int IH, int IW, int C, int OH,
                         int OW, int kh, int kw, int stride, int pad, int dil, int deform_groups, int offmask_stride,
                         void* stream);
int loft_mdcn_sample_bwd(const void* x, const float* offmask, const void* dcol, float* dx, float* doffmask, int dtype, int B,
                         int IH, int IW, int C, int OH, int OW, int kh, int kw, int stride, int pad, int dil,
                         int deform_groups, int offmask_stride, void* workspace, void* stream);
int64_t loft_mdcn_bwd_workspace_bytes(int B, int C, int OH, int OW, int kh, int kw, int stride, int dil, int deform_groups,
                                      int offmask_stride);

/* ---- weight fold + pack -----------------------------------------------------------------------------
 * Per conv and step: fp32 master weight [Cout][Cin][R][S] (the reference/checkpoint layout) -> bf16 operand packings
 * wp_fwd [R*S][Cout][Cin] and wp_dgrad [R*S][Cin][Cout] (either may be NULL) and the fp32 epilogue bias, folding the
 * frozen-statistics BatchNorm that follows the conv (norm_eval=True, mmdet/models/backbones/resnet.py:640-649; formula of
 * tools/fuse_conv_bn.py:10-23) when gamma != NULL; conv_bias is used when there is no BN.  loft_fold_unpack_bwd is its
 * chain rule: dwp fp32 [R*S][Cout][Cin] (gradient of the folded weight), db fp32 [Cout] (gradient of the folded bias)
 * -> dw [Cout][Cin][R][S], dgamma, dbeta (any of them may be NULL).  pack_f32 != 0 writes fp32 packings (parity mode).
 * CoutP >= Cout, CinP >= Cin: channel-padded packings [R*S][CoutP][CinP] / [R*S][CinP][CoutP] / bias [CoutP] with zeros in
 * the padding (HRNet's 32-channel branch is carried in 64-channel tensors whose upper half stays zero).
 * loft_fold_unpack_bwd accumulate != 0: dw / dgamma / dbeta are ADDED to (the trainer passes the parameters' slots of its
 * flat gradient arena, so no separate per-parameter accumulation launch is needed). */
int loft_fold_pack(const float* w, const float* conv_bias, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, int Cout, int Cin, int RS, void* wp_fwd, void* wp_dgrad, float* bias_out,
                   int pack_f32, int CoutP, int CinP, void* stream);
/* loft_fold_pack_multi: the bf16 packings of MANY convs in one launch.  desc (device): n records of 16 int64 {w, conv_bias,
 * gamma, beta, mean, var, wp_fwd, wp_dgrad, bias_out (device addresses, 0 = absent), eps as float bits, Cout, Cin, RS, CoutP,
 * CinP, first_chunk}; record i covers chunks [first_chunk_i, first_chunk_{i+1}); nchunks = their total.  A record with RS <= 9
 * has ceil(CoutP / NT) * ceil(CinP / 64) chunks (NT = 64 when RS == 1, else 16; chunk = one channel tile, all taps), a record
 * with more taps ceil(CoutP * CinP * RS / 2048).  CoutP and CinP must be even.  RS < 0 selects the "n-major" forward
 * packing wp[n][t][c] with |RS| taps (a Linear over a flattened [C,H,W] map whose activation is kept NHWC; such records
 * have Cout chunks, one output channel each, Cin * |RS| <= 18432, and their wp_dgrad is NOT written: the caller
 * derives it with loft_transpose_bf16); loft_fold_unpack_bwd_multi reads dwp in the same [n][t][c] order for RS < 0. */
int loft_fold_pack_multi(const int64_t* desc, int n, int64_t nchunks, void* stream);
/* The fp32 parity mode's per-step weight work in two launches (mmdet/core/fp16/hooks.py has no counterpart: the reference's fp32
 * path multiplies fp32 weights directly; here every fp32 operand is split into 16-bit planes).  loft_fold_f32_multi: loft_fold_pack
 * with fp32 packings for n records of 18 int64 {w, conv_bias, gamma, beta, mean, var, wp_fwd, wp_dgrad, bias_out, eps (float bits),
 * Cout, Cin, RS, CoutP, CinP, first_block, amax slot | 0, 0} (a block = 1024 elements of the padded forward packing; first_block
 * ascending); the absmax of each record's folded weights is max-folded into its PRE-ZEROED slot.  loft_split_planes_f32_multi:
 * loft_split_planes_f32 for n records of 6 int64 {src, planes, plane stride in elements, count (% 8 == 0), amax slot | 0, first_block}
 * (a block = 2048 elements). */
int loft_fold_f32_multi(const int64_t* desc, int n, int64_t nblocks, void* stream);
int loft_split_planes_f32_multi(const int64_t* desc, int n, int64_t nblocks, void* stream);
/* loft_fold_unpack_bwd_multi: loft_fold_unpack_bwd for MANY convs in one launch, accumulate-only (the trainer's direct gradient
 * sink).  desc (device): njobs records of 16 int64 {dwp, db, w, gamma, mean, var, dw, dgamma, dbeta_or_dbias (device addresses,
 * 0 = absent), eps as float bits, Cout, Cin, RS, CoutP, CinP, first_block}; record i owns blocks [first_block_i, first_block_i +
 * Cout_i); nblocks = their total.  lds_floats: dynamic LDS in floats (<= 16384): >= Cin * |RS| of every RS < 0 record
 * (required); records with RS > 1 and Cin * RS <= lds_floats interleave their taps through it (coalesced reads). */
int loft_fold_unpack_bwd_multi(const int64_t* desc, int njobs, int64_t nblocks, int lds_floats, void* stream);
/* loft_transpose_bf16: dst[Cc][R] = src[R][Cc]^T (row-major bf16, R and Cc even): the [K][O] data-gradient operand of an
 * n-major record from its [O][K] forward packing. */
int loft_transpose_bf16(const void* src, void* dst, int R, int Cc, void* stream);
int loft_fold_unpack_bwd(const float* dwp, const float* db, const float* w, const float* gamma, const float* mean,
                         const float* var, float eps, int Cout, int Cin, int RS, float* dw, float* dgamma, float* dbeta,
                         int CoutP, int CinP, int accumulate, void* stream);

/* loft_narrow_head_bwd: backward of a 1x1 conv / linear layer with Cout <= 8 outputs (FCNMaskHead.conv_logits,
 * fcn_mask_head.py:102; fc_cls / fc_reg, bbox_head.py:46-56; fc_offset, offset_head_expand_feature.py:104) in one pass:
 * gx[M,Cin] bf16 = (relu_in ? x > 0 : 1) * g[M,:Cout] . w[Cout,Cin];  dw[Cout,Cin] += g^T x and db[Cout] += colsum(g) with fp32
 * atomics (caller zeroes them).  g fp32 with row stride g_stride >= Cout, x bf16 [M,Cin], Cin % 4 == 0, Cin <= 1024.
 * gx / dw / db may be NULL. */
int loft_narrow_head_bwd(const float* g, int g_stride, const void* x, const float* w, int64_t M, int Cin, int Cout,
                         int relu_in, void* gx, float* dw, float* db, void* stream);
/* the same pass on fp32 activations (x, gx fp32 [M][Cin]; 16-byte accesses): the narrow heads of the fp32 parity mode (round 6) */
int loft_narrow_head_bwd_f32(const float* g, int g_stride, const float* x, const float* w, int64_t M, int Cin, int Cout,
                             int relu_in, float* gx, float* dw, float* db, void* stream);

/* loft_random_sample: RandomSampler.sample for a batch (mmdet/core/bbox/samplers/random_sampler.py:31-75,
 * base_sampler.py:34-101).  gt_inds int64 [B,N] (>0 positive, 0 negative, <0 ignored).  Per image: min(#pos, max_pos) positives,
 * then min(#neg, num - sampled_pos) negatives; mode 1 = uniformly random subsets (hashed keys, deterministic in `seed`), mode 0 =
 * the first ones in index order.  Outputs in ascending index order: pos_idx int64 [B,P] / pos_valid u8 [B,P] with
 * P = min(max_pos, N), neg_idx / neg_valid [B,Q] with Q = min(num, N); unused slots hold index N-1 and valid 0.
 * workspace: loft_random_sample_workspace_bytes(B, N) bytes (16-byte aligned) for the per-box class codes. */
int64_t loft_random_sample_workspace_bytes(int B, int N);
int loft_random_sample(const int64_t* gt_inds, int B, int N, int num, int max_pos, int mode, uint64_t seed,
                       int64_t* pos_idx, uint8_t* pos_valid, int64_t* neg_idx, uint8_t* neg_valid, void* workspace, void* stream);

/* loft_rpn_sample_gather: the RPN's target / prediction gathering for the sampled anchors of a batch
 * (anchor_head.py:187-237 targets, :429-497 loss inputs; rpn_head.py:38-54 outputs).  heads[l]: fused head output of level l,
 * fp32 [B,H_l,W_l,Cp] with channel a < A = objectness logit of anchor slot a, channel A + 4a + j = delta j; lvl_off[L+1]: first
 * flat anchor index of each level (flat index = lvl_off[l] + (y*W_l + x)*A + a).  For s < P the sample is pos_idx[b,s], else
 * neg_idx[b,s-P] (loft_random_sample outputs).  Writes vals [B,S,5] (logit, 4 deltas), rows int32 [B*S,4] = (b, level or -1 if
 * the slot is unused, y, x), slot int64 [B*S], tgt [B,P,4] = bbox2delta(anchor, gts[b, gt_inds-1]) (0 for unused slots), label
 * int64 [B,S] (1 = positive), weight [B,S] (1 = used).  S = P + Q. */
int loft_rpn_sample_gather(const void* const* heads, const int* H, const int* W, const int64_t* lvl_off, int num_levels,
                           int B, int Cp, int A, const float* anchors, const float* gts, int Kmax,
                           const int64_t* gt_inds, int64_t N, const int64_t* pos_idx, const uint8_t* pos_valid, int P,
                           const int64_t* neg_idx, const uint8_t* neg_valid, int Q, const float* means_host,
                           const float* stds_host, float* vals, int32_t* rows, int64_t* slot, float* tgt,
                           int64_t* label, float* weight, void* stream);

/* loft_sampled_avg_factor: out[0] = sum_b max(#pos_valid[b,:], 1) + sum_b max(#neg_valid[b,:], 1): the RPN losses' normaliser
 * `num_total_samples` of anchor_head.py:363-364, 462-464 from the sampler's validity bytes ([B,P] / [B,Q]); one launch. */
int loft_sampled_avg_factor(const uint8_t* pos_valid, const uint8_t* neg_valid, int B, int P, int Q, float* out, void* stream);

/* loft_fused_loss: a weighted, normalised loss and its gradient in one launch (mmdet/models/losses/utils.py:26-52 around
 * smooth_l1_loss.py:8-50 and cross_entropy_loss.py:9-125).  mode 0 L1, 1 SmoothL1(beta), 2 sigmoid cross-entropy on logits (target
 * fp32 in [0,1]), 3 softmax cross-entropy (pred [n,C], target int64 [n], weight per row).  n = number of elements (rows for mode 3);
 * weight fp32 per element / row or NULL.  loss_out[0] = scale / denom * sum_i w_i l_i, grad (same shape as pred) = d loss / d pred;
 * denom = *avg_factor (device scalar) if not NULL else `count`.  partial: >= 256 floats of scratch; counter: one uint32 that is 0
 * before the first launch (the kernel leaves it 0).  Deterministic summation order. */
int loft_fused_loss(int mode, const float* pred, const void* target, const float* weight, int64_t n, int C,
                    const float* avg_factor, float count, float scale, float beta, float* grad, float* partial,
                    uint32_t* counter, float* loss_out, void* stream);
/* loft_fused_loss_v2: the same launch reading its operands where the heads left them (no clone / cast / compare launches in front):
 * pred is a strided view -- logical index i (element; row for mode 3, whose C classes are read at unit stride) sits at
 * (i / (d1 d2)) s0 + ((i / d2) % d1) s1 + (i % d2) s2 floats from `pred` (contiguous: d1 = d2 = 1, s0 = 1, or C for mode 3);
 * target_kind 1 (mode 2 only): int64 labels, target = (label >= 1) (cross_entropy_loss.py:60-66); weight_kind 1: uint8 weights;
 * every `wdiv` consecutive logical elements share one weight.  want_acc (mode 3 only): loss_out[1] = top-1 accuracy in percent
 * (accuracy.py:4-48), partial then needs >= 512 floats.  grad is dense in logical order. */
int loft_fused_loss_v2(int mode, const float* pred, int64_t d1, int64_t d2, int64_t s0, int64_t s1, int64_t s2, const void* target,
                       int target_kind, const void* weight, int weight_kind, int64_t wdiv, int64_t n, int C, const float* avg_factor,
                       float count, float scale, float beta, float* grad, float* partial, uint32_t* counter, float* loss_out,
                       int want_acc, void* stream);

/* loft_roi_sample_targets: SamplingResult + bbox2roi + BBoxHead.get_targets for a batch (sampling_result.py:25-53,
 * transforms.py:54-73, bbox_head.py:84-138).  cand [B,Ncand,4] candidate boxes (gts first when add_gt_as_proposals), gt_inds int64
 * [B,Ncand], gts [B,Kmax,4], gt_labels int64 [B,Kmax]; pos_idx [B,P] / neg_idx [B,Q] from loft_random_sample (valid slots first);
 * npos / nneg int32 [B] = numbers of valid slots, roi_off / pos_off int32 [B] = exclusive prefix sums of npos+nneg / npos (device
 * arrays built from ONE host read of the counts).  Writes, per image [pos..., neg...]: rois [M,5] (b, x1, y1, x2, y2), labels int64
 * [M] (num_classes for negatives), label_weights [M] = 1, bbox_targets / bbox_weights [M,4] (bbox2delta for positives, weight 1),
 * and the positives' lists pos_rois [Np,5], pos_img / pos_gt / pos_row int64 [Np] (image, assigned gt, row in rois). */
/* The tables loft_roi_sample_targets reads, computed on the device from the sampler's validity flags (uint8 [B,P] / [B,Q]):
 * tab int32 [4][B] = {positives, negatives, first RoI row, first positive row} per image, B <= 64.  Lets the caller launch
 * loft_roi_sample_targets (outputs sized for the worst case) without first reading the counts on the host. */
int loft_roi_sample_offsets(const uint8_t* pos_valid, const uint8_t* neg_valid, int B, int P, int Q, int32_t* tab, void* stream);
int loft_roi_sample_targets(const float* cand, int Ncand, const int64_t* gt_inds, const float* gts,
                            const int64_t* gt_labels, int Kmax, const int64_t* pos_idx, const int64_t* neg_idx, int P,
                            int Q, int B, const int32_t* npos_dev, const int32_t* nneg_dev, const int32_t* roi_off_dev,
                            const int32_t* pos_off_dev, int num_classes, const float* means_host, const float* stds_host,
                            float* rois, int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
                            float* pos_rois, int64_t* pos_img, int64_t* pos_gt, int64_t* pos_row, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOFT_HIP_H */
