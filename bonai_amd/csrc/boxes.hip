// boxes.hip -- box / target arithmetic of the LOFT path on gfx950 (fp32 + integer, HBM/latency bound).
// Compiled with -ffp-contract=off: IoUs and thresholds round exactly like the CPU restatement
// (oracle/ops_ref.py), so assignment indices are bit-exact.
//
//   iou_assign       MaxIoUAssigner.assign incl. the per-gt low-quality loop, without materialising
//                    the KxN IoU matrix (mmdet/core/bbox/assigners/max_iou_assigner.py:60-212,
//                    iou_calculators/iou2d_calculator.py:39-130)
//   rpn_scores/rpn_decode  sigmoid + gather + delta2bbox of the sorted top-k candidates
//                    (mmdet/models/dense_heads/rpn_head.py:116-150, delta_xywh_bbox_coder.py:119-197)
//   delta2bbox / bbox2delta   DeltaXYWHBBoxCoder (delta_xywh_bbox_coder.py:78-197)
//   foa_targets      4-rotation offset targets (offset_head_expand_feature.py:271-344,
//                    delta_xy_offset_coder.py:46-65) -- replaces the per-RoI python loop
//   foa_fuse_decode  offset_fusion('max') + DeltaXYOffsetCoder.decode (offset_head_expand_feature.py:346-448)
//   mask_target      BitmapMasks.crop_and_resize via RoIAlign(28, aligned) >= 0.5 on device
//                    (mmdet/core/mask/mask_target.py:33-62, structures.py:261-291) -- replaces the
//                    GPU->CPU->GPU round trip
#include "loft_common.h"
#include <algorithm>
#include "../../include/loft_hip.h"

__device__ __forceinline__ float iou_pair(const float4 g, const float4 b) {
    // bboxes1 = gt (g), bboxes2 = box (b); iou2d_calculator.py:110-128
    const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
    const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float overlap = w * h;
    const float a1 = (g.z - g.x) * (g.w - g.y);
    const float a2 = (b.z - b.x) * (b.w - b.y);
    const float uni = fmaxf(a1 + a2 - overlap, 1e-6f);
    return overlap / uni;
}

// Bounding box of the wavefront's 64 boxes (scalar registers).  Anchors are enumerated position-major, so a wavefront of the RPN
// assignment covers a strip of ~21 grid positions and intersects 2-5 of an image's 80 gt boxes: a gt outside the strip has
// IoU == 0 with every lane (w or h clamps to 0), cannot raise a maximum (all maxima start at 0 with strict '>' updates, the
// first-index tie rule of torch.max is kept) and cannot equal a gt maximum >= min_pos_iou > 0, so it is skipped outright.
struct WaveBox { float x1, y1, x2, y2; };
__device__ __forceinline__ WaveBox wave_bbox(const float4 bx, bool live) {
    float x1 = live ? bx.x : 3.4e38f, y1 = live ? bx.y : 3.4e38f, x2 = live ? bx.z : -3.4e38f, y2 = live ? bx.w : -3.4e38f;
    for (int o = 32; o > 0; o >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, o, 64)); y1 = fminf(y1, __shfl_xor(y1, o, 64));
        x2 = fmaxf(x2, __shfl_xor(x2, o, 64)); y2 = fmaxf(y2, __shfl_xor(y2, o, 64));
    }
    WaveBox w;
    w.x1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x1)));
    w.y1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y1)));
    w.x2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x2)));
    w.y2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y2)));
    return w;
}
__device__ __forceinline__ bool gt_outside(const float4 g, const WaveBox& w) {
    return (g.z <= w.x1) | (g.x >= w.x2) | (g.w <= w.y1) | (g.y >= w.y2);
}

// pass 1: per box max/argmax over gts; per gt max over boxes.  The per-gt maximum is reduced inside the
// wavefront (DPP/shuffle max), then across the block's 4 waves in LDS, and only then published with one
// atomicMax per (block, gt) on the non-negative float's bit pattern -- not one per (box, gt).
__global__ __launch_bounds__(256) void iou_pass1_kernel(const float* __restrict__ boxes, const int* __restrict__ nbox, int Nmax,
                                                        const float* __restrict__ gts, const int* __restrict__ ngt, int Kmax,
                                                        float* __restrict__ max_ov, int32_t* __restrict__ argmax,
                                                        unsigned* __restrict__ gt_max_bits) {
    extern __shared__ float4 sg[];
    unsigned* sgm = reinterpret_cast<unsigned*>(sg + Kmax);
    const int b = blockIdx.y;
    const int K = ngt[b], N = nbox[b];
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        sg[i] = reinterpret_cast<const float4*>(gts)[(long)b * Kmax + i];
        sgm[i] = 0u;
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n < N;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) bx = reinterpret_cast<const float4*>(boxes)[(long)b * Nmax + n];
    const WaveBox wb = wave_bbox(bx, live);
    float best = 0.f;
    int bi = 0;
    for (int i = 0; i < K; ++i) {
        const float4 g = sg[i];
        if (gt_outside(g, wb)) continue;                  // wave-uniform (scalar) branch
        float v = live ? iou_pair(g, bx) : 0.f;
        if (v > best) { best = v; bi = i; }
        float m = v;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(sgm + i, __float_as_uint(m));
    }
    if (live) {
        max_ov[(long)b * Nmax + n] = best;
        argmax[(long)b * Nmax + n] = bi;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x)
        if (sgm[i] != 0u) atomicMax(gt_max_bits + (long)b * Kmax + i, sgm[i]);
}

__global__ __launch_bounds__(256) void iou_pass2_kernel(const float* __restrict__ boxes, const int* __restrict__ nbox, int Nmax,
                                                        const float* __restrict__ gts, const int* __restrict__ ngt, int Kmax,
                                                        const float* __restrict__ max_ov, const int32_t* __restrict__ argmax,
                                                        const unsigned* __restrict__ gt_max_bits, float pos_thr, float neg_thr,
                                                        float min_pos, int low_quality, int64_t* __restrict__ gt_inds) {
    extern __shared__ float4 sg[];
    float* sgm = reinterpret_cast<float*>(sg + Kmax);
    const int b = blockIdx.y;
    const int K = ngt[b], N = nbox[b];
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        sg[i] = reinterpret_cast<const float4*>(gts)[(long)b * Kmax + i];
        sgm[i] = __uint_as_float(gt_max_bits[(long)b * Kmax + i]);
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n < N;
    long a = -1;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && K > 0) {
        const float mo = max_ov[(long)b * Nmax + n];
        if (mo >= 0.f && mo < neg_thr) a = 0;
        if (mo >= pos_thr) a = argmax[(long)b * Nmax + n] + 1;
        bx = reinterpret_cast<const float4*>(boxes)[(long)b * Nmax + n];
    } else if (live) {
        a = 0;
    }
    if (low_quality && K > 0 && min_pos > 0.f) {
        const WaveBox wb = wave_bbox(bx, live);
        for (int i = 0; i < K; ++i) {
            const float gm = sgm[i];
            const float4 g = sg[i];
            if (!(gm >= min_pos) || gt_outside(g, wb)) continue;      // wave-uniform
            if (live && iou_pair(g, bx) == gm) a = i + 1;
        }
    } else if (low_quality && K > 0) {
        for (int i = 0; i < K; ++i) {
            const float gm = sgm[i];
            if (live && gm >= min_pos && iou_pair(sg[i], bx) == gm) a = i + 1;
        }
    }
    if (n < Nmax) gt_inds[(long)b * Nmax + n] = a;
}

LOFT_EXPORT int loft_iou_assign(const float* boxes, const int* nbox, int Nmax, const float* gts, const int* ngt, int Kmax,
                                int B, float pos_thr, float neg_thr, float min_pos, int low_quality, float* max_ov,
                                int32_t* argmax_ws, uint32_t* gt_max_ws, int64_t* gt_inds, void* stream) {
    if (B <= 0 || Nmax <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gt_max_ws, 0, sizeof(uint32_t) * (size_t)B * (Kmax > 0 ? Kmax : 1), s);
    if (e != hipSuccess) return (int)e;
    dim3 grid(loft_cdiv(Nmax, 256), B);
    const size_t sh = (size_t)(Kmax > 0 ? Kmax : 1) * (sizeof(float4) + sizeof(float));
    hipLaunchKernelGGL(iou_pass1_kernel, grid, dim3(256), sh, s, boxes, nbox, Nmax, gts, ngt, Kmax, max_ov, argmax_ws,
                       gt_max_ws);
    LOFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(iou_pass2_kernel, grid, dim3(256), sh, s, boxes, nbox, Nmax, gts, ngt, Kmax, max_ov, argmax_ws,
                       gt_max_ws, pos_thr, neg_thr, min_pos, low_quality, gt_inds);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- DeltaXYWHBBoxCoder ---------------------------------------------------------------------
__device__ __forceinline__ float4 decode_box(const float4 r, float d0, float d1, float d2, float d3, const float* means,
                                             const float* stds, float max_ratio, float max_h, float max_w) {
    const float dx = d0 * stds[0] + means[0], dy = d1 * stds[1] + means[1];
    float dw = d2 * stds[2] + means[2], dh = d3 * stds[3] + means[3];
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f;
    const float pw = r.z - r.x, ph = r.w - r.y;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dx, gy = py + ph * dy;
    float4 o;
    o.x = gx - gw * 0.5f; o.y = gy - gh * 0.5f; o.z = gx + gw * 0.5f; o.w = gy + gh * 0.5f;
    if (max_w > 0.f) {
        o.x = fminf(fmaxf(o.x, 0.f), max_w); o.z = fminf(fmaxf(o.z, 0.f), max_w);
        o.y = fminf(fmaxf(o.y, 0.f), max_h); o.w = fminf(fmaxf(o.w, 0.f), max_h);
    }
    return o;
}

struct Coder4 { float means[4], stds[4]; };

__global__ void delta2bbox_kernel(const float* __restrict__ rois, const float* __restrict__ deltas, long n, Coder4 c,
                                  float max_ratio, float max_h, float max_w, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = reinterpret_cast<const float4*>(rois)[i];
    const float4 d = reinterpret_cast<const float4*>(deltas)[i];
    reinterpret_cast<float4*>(out)[i] = decode_box(r, d.x, d.y, d.z, d.w, c.means, c.stds, max_ratio, max_h, max_w);
}
LOFT_EXPORT int loft_delta2bbox(const float* rois, const float* deltas, int64_t n, const float* means_host,
                                const float* stds_host, float wh_ratio_clip, float max_h, float max_w, float* out,
                                void* stream) {
    if (n <= 0) return 0;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    const float max_ratio = fabsf(logf(wh_ratio_clip));
    hipLaunchKernelGGL(delta2bbox_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, deltas, (long)n, c,
                       max_ratio, max_h, max_w, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

__global__ void bbox2delta_kernel(const float* __restrict__ props, const float* __restrict__ gt, long n, Coder4 c,
                                  float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = reinterpret_cast<const float4*>(props)[i];
    const float4 g = reinterpret_cast<const float4*>(gt)[i];
    const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f, pw = p.z - p.x, ph = p.w - p.y;
    const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
    float4 o;
    o.x = ((gx - px) / pw - c.means[0]) / c.stds[0];
    o.y = ((gy - py) / ph - c.means[1]) / c.stds[1];
    o.z = (logf(gw / pw) - c.means[2]) / c.stds[2];
    o.w = (logf(gh / ph) - c.means[3]) / c.stds[3];
    reinterpret_cast<float4*>(out)[i] = o;
}
LOFT_EXPORT int loft_bbox2delta(const float* proposals, const float* gt, int64_t n, const float* means_host,
                                const float* stds_host, float* out, void* stream) {
    if (n <= 0) return 0;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    hipLaunchKernelGGL(bbox2delta_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, proposals, gt, (long)n, c,
                       out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- RPN proposal candidates -----------------------------------------------------------------
// head output of one level: fp32 [B,H,W,Cp]; channel a (<A) = objectness logit of anchor a,
// channel A + 4a + j = delta j of anchor a.  keys[b][lvl_off + pos*A + a] = sigmoid(logit).
__global__ void rpn_scores_kernel(const float* __restrict__ head, int B, int HW, int Cp, int A, long img_stride, long lvl_off,
                                  float* __restrict__ keys) {
    const long n = (long)B * HW * A;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = (int)(i % A);
    const long p = i / A;
    const int pos = (int)(p % HW);
    const int b = (int)(p / HW);
    const float x = head[((long)b * HW + pos) * Cp + a];
    keys[(long)b * img_stride + lvl_off + (long)pos * A + a] = 1.f / (1.f + expf(-x));
}
LOFT_EXPORT int loft_rpn_scores(const float* head, int B, int H, int W, int Cp, int A, int64_t img_stride, int64_t lvl_off,
                                float* keys, void* stream) {
    const long n = (long)B * H * W * A;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(rpn_scores_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, head, B, H * W, Cp, A,
                       (long)img_stride, (long)lvl_off, keys);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// For level `lvl` of every image: rank r < topk of the sorted candidate list -> anchor + deltas -> box.
// sorted_idx holds, per (image, level) segment, the within-image candidate index (lvl_off + pos*A + a).
// out_boxes [B][cand_stride][4] at slot cand_off + r.
__global__ void rpn_decode_kernel(const float* __restrict__ head, const int32_t* __restrict__ sorted_idx, int B, int H, int W,
                                  int Cp, int A, long img_stride, long lvl_off, int topk, const float* __restrict__ base_anchors,
                                  int stride, Coder4 c, float max_ratio, float max_h, float max_w, long cand_stride, long cand_off,
                                  float* __restrict__ out_boxes) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * topk) return;
    const int r = (int)(i % topk), b = (int)(i / topk);
    const int idx = sorted_idx[(long)b * img_stride + lvl_off + r] - (int)lvl_off - (int)((long)b * img_stride);
    const int a = idx % A, pos = idx / A;
    const int y = pos / W, x = pos - y * W;
    const float sx = (float)(x * stride), sy = (float)(y * stride);
    float4 anc;
    anc.x = base_anchors[a * 4 + 0] + sx; anc.y = base_anchors[a * 4 + 1] + sy;
    anc.z = base_anchors[a * 4 + 2] + sx; anc.w = base_anchors[a * 4 + 3] + sy;
    const float* d = head + ((long)b * H * W + pos) * Cp + A + a * 4;
    reinterpret_cast<float4*>(out_boxes)[(long)b * cand_stride + cand_off + r] =
        decode_box(anc, d[0], d[1], d[2], d[3], c.means, c.stds, max_ratio, max_h, max_w);
}
LOFT_EXPORT int loft_rpn_decode(const float* head, const int32_t* sorted_idx, int B, int H, int W, int Cp, int A,
                                int64_t img_stride, int64_t lvl_off, int topk, const float* base_anchors, int stride,
                                const float* means_host, const float* stds_host, float wh_ratio_clip, float max_h, float max_w,
                                int64_t cand_stride, int64_t cand_off, float* out_boxes, void* stream) {
    const long n = (long)B * topk;
    if (n <= 0) return 0;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, head, sorted_idx, B, H, W,
                       Cp, A, (long)img_stride, (long)lvl_off, topk, base_anchors, stride, c, fabsf(logf(wh_ratio_clip)), max_h,
                       max_w, (long)cand_stride, (long)cand_off, out_boxes);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- the same two steps for ALL pyramid levels in one launch each (the training step's proposal chain: five + five launches of
// 5 us on a chain every RoI-head launch waits for), plus what followed them as separate small launches: the decode also writes the
// candidates' scores in the candidate layout and the per-image maximum coordinate batched_nms shifts the levels by
// (mmcv batched_nms: boxes + idx * (boxes.max() + 1)); the scores launch resets that maximum.
#define RPN_MAX_LEVELS 8
struct RpnLvl { const float* head; const float* base; int H, W, topk, stride; long lvl_off, cand_off; };
struct RpnLvls { int n; RpnLvl l[RPN_MAX_LEVELS]; };

__global__ void rpn_scores_levels_kernel(const RpnLvls lv, int B, int Cp, int A, long img_stride, float* __restrict__ keys,
                                         float* __restrict__ img_max) {
    const RpnLvl& L = lv.l[blockIdx.y];
    const int HW = L.H * L.W;
    const long n = (long)B * HW * A;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (img_max != nullptr && blockIdx.y == 0 && i < B) img_max[i] = -INFINITY;
    if (i >= n) return;
    const int a = (int)(i % A);
    const long p = i / A;
    const int pos = (int)(p % HW);
    const int b = (int)(p / HW);
    const float x = L.head[((long)b * HW + pos) * Cp + a];
    keys[(long)b * img_stride + L.lvl_off + (long)pos * A + a] = 1.f / (1.f + expf(-x));
}

// any sign; *addr starts at -inf.  Same-address read-modify-writes serialise in L2 (~0.1 us each: 1600 wave maxima onto 8 addresses
// made the decode launch 42 us); a maximum only grows, so a value not above what a plain L2 read returns is dropped -- after the
// first few boxes that touch the clipping border that is every one.
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (!(v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

__global__ void rpn_decode_levels_kernel(const RpnLvls lv, const int32_t* __restrict__ sorted_idx, const float* __restrict__ sorted_keys,
                                         int B, int Cp, int A, long img_stride, Coder4 c, float max_ratio, float max_h, float max_w,
                                         long cand_stride, float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                         float* __restrict__ img_max) {
    const RpnLvl& L = lv.l[blockIdx.y];
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const bool in = i < (long)B * L.topk;
    int b = -1;
    float m = -INFINITY;
    if (in) {
        const int r = (int)(i % L.topk);
        b = (int)(i / L.topk);
        const long src = (long)b * img_stride + L.lvl_off + r;
        const int idx = sorted_idx[src] - (int)L.lvl_off - (int)((long)b * img_stride);
        const int a = idx % A, pos = idx / A;
        const int y = pos / L.W, x = pos - y * L.W;
        const float sx = (float)(x * L.stride), sy = (float)(y * L.stride);
        float4 anc;
        anc.x = L.base[a * 4 + 0] + sx; anc.y = L.base[a * 4 + 1] + sy;
        anc.z = L.base[a * 4 + 2] + sx; anc.w = L.base[a * 4 + 3] + sy;
        const float* d = L.head + ((long)b * L.H * L.W + pos) * Cp + A + a * 4;
        const float4 box = decode_box(anc, d[0], d[1], d[2], d[3], c.means, c.stds, max_ratio, max_h, max_w);
        const long dst = (long)b * cand_stride + L.cand_off + r;
        reinterpret_cast<float4*>(out_boxes)[dst] = box;
        if (out_scores != nullptr) out_scores[dst] = sorted_keys[src];
        m = fmaxf(fmaxf(box.x, box.y), fmaxf(box.z, box.w));
    }
    if (img_max == nullptr) return;
    // one atomic per wave when the wave sits inside one image (all but the waves that straddle an image boundary)
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    if (__all(b == b0 || !in) && b0 >= 0) {
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) atomic_max_float(img_max + b0, m);
    } else if (in) {
        atomic_max_float(img_max + b, m);
    }
}

static int rpn_levels_fill(RpnLvls& lv, const void* const* heads, const void* const* bases, const int* H, const int* W,
                           const int* topk, const int* stride, const int64_t* lvl_off, const int64_t* cand_off, int levels) {
    if (levels < 1 || levels > RPN_MAX_LEVELS) return (int)hipErrorInvalidValue;
    lv.n = levels;
    for (int l = 0; l < levels; ++l) {
        lv.l[l].head = (const float*)heads[l];
        lv.l[l].base = bases ? (const float*)bases[l] : nullptr;
        lv.l[l].H = H[l]; lv.l[l].W = W[l];
        lv.l[l].topk = topk ? topk[l] : 0;
        lv.l[l].stride = stride ? stride[l] : 0;
        lv.l[l].lvl_off = (long)lvl_off[l];
        lv.l[l].cand_off = cand_off ? (long)cand_off[l] : 0l;
    }
    return 0;
}

LOFT_EXPORT int loft_rpn_scores_levels(const void* const* heads_host, const int* H_host, const int* W_host,
                                       const int64_t* lvl_off_host, int levels, int B, int Cp, int A, int64_t img_stride,
                                       float* keys, float* img_max, void* stream) {
    RpnLvls lv;
    if (int e = rpn_levels_fill(lv, heads_host, nullptr, H_host, W_host, nullptr, nullptr, lvl_off_host, nullptr, levels)) return e;
    long nmax = B;
    for (int l = 0; l < levels; ++l) nmax = std::max(nmax, (long)B * H_host[l] * W_host[l] * A);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(rpn_scores_levels_kernel, dim3(loft_cdiv(nmax, 256), levels), dim3(256), 0, (hipStream_t)stream, lv, B, Cp, A,
                       (long)img_stride, keys, img_max);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_rpn_decode_levels(const void* const* heads_host, const void* const* base_anchors_host, const int* H_host,
                                       const int* W_host, const int* topk_host, const int* stride_host, const int64_t* lvl_off_host,
                                       const int64_t* cand_off_host, int levels, const int32_t* sorted_idx, const float* sorted_keys,
                                       int B, int Cp, int A, int64_t img_stride, const float* means_host, const float* stds_host,
                                       float wh_ratio_clip, float max_h, float max_w, int64_t cand_stride, float* out_boxes,
                                       float* out_scores, float* img_max, void* stream) {
    RpnLvls lv;
    if (int e = rpn_levels_fill(lv, heads_host, base_anchors_host, H_host, W_host, topk_host, stride_host, lvl_off_host, cand_off_host,
                                levels)) return e;
    long nmax = 0;
    for (int l = 0; l < levels; ++l) nmax = std::max(nmax, (long)B * topk_host[l]);
    if (nmax <= 0) return 0;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    hipLaunchKernelGGL(rpn_decode_levels_kernel, dim3(loft_cdiv(nmax, 256), levels), dim3(256), 0, (hipStream_t)stream, lv, sorted_idx,
                       sorted_keys, B, Cp, A, (long)img_stride, c, fabsf(logf(wh_ratio_clip)), max_h, max_w, (long)cand_stride,
                       out_boxes, out_scores, img_max);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// The end of the proposal chain (rpn_head.py:169-171 `dets[:cfg.nms_post]` per image + what the RoI head wants of it): the first
// `post` entries of every image's score-sorted survivor list -> props [B][post][5] (box, score; rows past the survivors zero)
// and counts [B].  top_scores / top_idx are loft_segmented_topk_desc's outputs over segments of `seg_stride` candidates
// (suppressed candidates carry -1); top_idx indexes cand_boxes [B * seg_stride][4].  One workgroup per image.
__global__ __launch_bounds__(256) void rpn_finalize_kernel(const float* __restrict__ top_scores, const int32_t* __restrict__ top_idx,
                                                           const float* __restrict__ cand_boxes, long seg_stride, int post,
                                                           float* __restrict__ props, int64_t* __restrict__ counts) {
    __shared__ int wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int cnt = 0;
    for (int r = tid; r < post; r += 256) {
        const float sc = top_scores[(long)b * seg_stride + r];
        const bool valid = sc >= 0.f;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) box = reinterpret_cast<const float4*>(cand_boxes)[top_idx[(long)b * seg_stride + r]];
        float* o = props + ((long)b * post + r) * 5;
        o[0] = box.x; o[1] = box.y; o[2] = box.z; o[3] = box.w; o[4] = valid ? sc : 0.f;
        cnt += valid ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) counts[b] = (int64_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}
LOFT_EXPORT int loft_rpn_finalize(const float* top_scores, const int32_t* top_idx, const float* cand_boxes, int B,
                                  int64_t seg_stride, int post, float* props, int64_t* counts, void* stream) {
    if (B <= 0) return 0;
    if (post < 0 || post > seg_stride) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rpn_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, top_scores, top_idx, cand_boxes,
                       (long)seg_stride, post, props, counts);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- FOA targets ---------------------------------------------------------------------------------
// out[k*N + i] = target of RoI i in rotation branch k (k = 0..3 <-> 0/90/180/270 deg).
// The reference rotates the gt offset by -k*90 deg through a python-float polar round trip; in fp32
// that equals the exact permutation (x,y),(y,-x),(-x,-y),(-y,x), then encodes
// (gx/pw, gy/ph)/std, with x' normalised by ph and y' by pw for the 90/270 branches (appendix A.5).
__global__ void foa_targets_kernel(const float* __restrict__ pos_boxes, const float* __restrict__ gt_off, long n, float std_x,
                                   float std_y, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = reinterpret_cast<const float4*>(pos_boxes)[i];
    const float ox = gt_off[2 * i], oy = gt_off[2 * i + 1];
    const float pw = p.z - p.x, ph = p.w - p.y;
    const float rx[4] = {ox, oy, -ox, -oy};
    const float ry[4] = {oy, -ox, -oy, ox};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float tx, ty;
        if (k & 1) {  // encode(pos, (y',x')) then swap back: x' / ph / std_y... see offset_head_expand_feature.py:295-298
            const float e0 = (ry[k] / pw) / std_x;  // encoded "x" slot holds y'
            const float e1 = (rx[k] / ph) / std_y;  // encoded "y" slot holds x'
            tx = e1; ty = e0;
        } else {
            tx = (rx[k] / pw) / std_x;
            ty = (ry[k] / ph) / std_y;
        }
        out[((long)k * n + i) * 2 + 0] = tx;
        out[((long)k * n + i) * 2 + 1] = ty;
    }
}
LOFT_EXPORT int loft_foa_targets(const float* pos_boxes, const float* pos_gt_offsets, int64_t n, float std_x, float std_y,
                                 float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(foa_targets_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos_boxes, pos_gt_offsets,
                       (long)n, std_x, std_y, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// pred [4N,2] branch-major -> fused+decoded offsets [N,2]
__global__ void foa_fuse_decode_kernel(const float* __restrict__ pred, const float* __restrict__ boxes, long n, float std_x,
                                       float std_y, float max_h, float max_w, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b0x = pred[(0 * n + i) * 2], b0y = pred[(0 * n + i) * 2 + 1];
    const float b1x = pred[(1 * n + i) * 2], b1y = pred[(1 * n + i) * 2 + 1];
    const float b2x = pred[(2 * n + i) * 2], b2y = pred[(2 * n + i) * 2 + 1];
    const float b3x = pred[(3 * n + i) * 2], b3y = pred[(3 * n + i) * 2 + 1];
    const float vx = fmaxf(fmaxf(fabsf(b0x), fabsf(b1y)), fmaxf(fabsf(b2x), fabsf(b3y)));
    const float vy = fmaxf(fmaxf(fabsf(b0y), fabsf(b1x)), fmaxf(fabsf(b2y), fabsf(b3x)));
    const float fx = vx * (b0x > 0.f ? 1.f : -1.f), fy = vy * (b0y > 0.f ? 1.f : -1.f);
    const float4 r = reinterpret_cast<const float4*>(boxes)[i];
    float gx = (r.z - r.x) * (fx * std_x), gy = (r.w - r.y) * (fy * std_y);
    gx = fminf(fmaxf(gx, -max_w), max_w);
    gy = fminf(fmaxf(gy, -max_h), max_h);
    out[2 * i] = gx; out[2 * i + 1] = gy;
}
LOFT_EXPORT int loft_foa_fuse_decode(const float* pred, const float* boxes, int64_t n, float std_x, float std_y, float max_h,
                                     float max_w, float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(foa_fuse_decode_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, boxes, (long)n,
                       std_x, std_y, max_h, max_w, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- plain LOFT OffsetHead (no FOA): targets and inference decode ------------------------------
// (attribute_heads/offset_head.py:118-188 get_targets / _offset_target_single, :190-243 get_offsets;
//  delta_xy_offset_coder.py:46-88).  reg_num = 2: out[i] = ((gx/pw, gy/ph) - mean)/std.  reg_num = 3 ("polar" heads):
//  the two encoded values are read as (length, angle) and the target is (length, cos(angle), sin(angle)) (:176-183).
__global__ void offset_targets_kernel(const float* __restrict__ pos_boxes, const float* __restrict__ gt_off, long n, float mean_x,
                                      float mean_y, float std_x, float std_y, int reg_num, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = reinterpret_cast<const float4*>(pos_boxes)[i];
    const float pw = p.z - p.x, ph = p.w - p.y;
    const float dx = (gt_off[2 * i] / pw - mean_x) / std_x;
    const float dy = (gt_off[2 * i + 1] / ph - mean_y) / std_y;
    if (reg_num == 2) { out[2 * i] = dx; out[2 * i + 1] = dy; }
    else { out[3 * i] = dx; out[3 * i + 1] = cosf(dy); out[3 * i + 2] = sinf(dy); }
}
LOFT_EXPORT int loft_offset_targets(const float* pos_boxes, const float* pos_gt_offsets, int64_t n, float mean_x, float mean_y,
                                    float std_x, float std_y, int reg_num, float* out, void* stream) {
    if (reg_num != 2 && reg_num != 3) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(offset_targets_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos_boxes, pos_gt_offsets,
                       (long)n, mean_x, mean_y, std_x, std_y, reg_num, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}
// pred [n, reg_num] -> offsets [n,2]: reg_num = 3 first folds (length, cos, sin) to (length, atan2(sin, cos)); decode
// (d*std + mean) * (pw, ph) clamped to +-(max_w, max_h); polar != 0 then maps (length, angle) -> length*(cos, sin) (:232-236).
__global__ void offset_decode_kernel(const float* __restrict__ pred, const float* __restrict__ boxes, long n, float mean_x,
                                     float mean_y, float std_x, float std_y, float max_h, float max_w, int reg_num, int polar,
                                     float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d0, d1;
    if (reg_num == 2) { d0 = pred[2 * i]; d1 = pred[2 * i + 1]; }
    else { d0 = pred[3 * i]; d1 = atan2f(pred[3 * i + 2], pred[3 * i + 1]); }
    const float4 r = reinterpret_cast<const float4*>(boxes)[i];
    float gx = (r.z - r.x) * (d0 * std_x + mean_x), gy = (r.w - r.y) * (d1 * std_y + mean_y);
    gx = fminf(fmaxf(gx, -max_w), max_w);
    gy = fminf(fmaxf(gy, -max_h), max_h);
    if (polar) { const float l = gx, a = gy; gx = l * cosf(a); gy = l * sinf(a); }
    out[2 * i] = gx; out[2 * i + 1] = gy;
}
LOFT_EXPORT int loft_offset_decode(const float* pred, const float* boxes, int64_t n, float mean_x, float mean_y, float std_x,
                                   float std_y, float max_h, float max_w, int reg_num, int polar, float* out, void* stream) {
    if (reg_num != 2 && reg_num != 3) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(offset_decode_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, boxes, (long)n,
                       mean_x, mean_y, std_x, std_y, max_h, max_w, reg_num, polar, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- polygon -> instance bitmap on the device ------------------------------------------------------------------------------
// Replaces LoadAnnotations._poly2mask (mmdet/datasets/pipelines/loading.py:301-326: pycocotools frPyObjects + merge + decode) so
// that the K x H x W uint8 masks of a tile are never built on the host nor uploaded: the polygons' vertices go up
// (a few KB) and each instance's bitmap is rasterised where mask_target reads it.  Algorithm = pycocotools' rleFrPoly restated
// ([pycocotools, not in tree]; oracle/ops_ref.py::poly2mask): vertices to a 5x grid, every edge walked along its major axis,
// one crossing (x, y) wherever the walk enters a new 5x column that is a pixel centre, the column-major run-length code
// toggles at x*H + y; decode = running parity.  One workgroup per INSTANCE (its polygons are OR-merged, = maskUtils.merge):
// the H*W+1 toggle bits live in LDS (128 KiB + 4 B for a 1024 x 1024 tile), crossings are XOR-ed in with LDS atomics, each
// thread then owns columns: parity carried word by word down the column, bytes written row-major (coalesced across the
// column-threads).  fp64 throughout like the C original; this file is built with -ffp-contract=off.
struct PolyPt { int u, v; };
__device__ __forceinline__ PolyPt poly_edge_point(const double* xy, int k, int e, int d) {
    const double scale = 5.0;
    const int e1 = (e + 1 == k) ? 0 : e + 1;
    int xs = (int)(scale * xy[2 * e] + .5), ys = (int)(scale * xy[2 * e + 1] + .5);
    int xe = (int)(scale * xy[2 * e1] + .5), ye = (int)(scale * xy[2 * e1 + 1] + .5);
    const int dx = abs(xe - xs), dy = abs(ys - ye);
    const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    PolyPt p;
    if (dx >= dy) {
        const double s = dx > 0 ? (double)(ye - ys) / dx : 0.0;
        const int t = flip ? dx - d : d;
        p.u = t + xs; p.v = (int)(ys + s * t + .5);
    } else {
        const double s = (double)(xe - xs) / dy;
        const int t = flip ? dy - d : d;
        p.v = t + ys; p.u = (int)(xs + s * t + .5);
    }
    return p;
}

__global__ __launch_bounds__(1024) void poly2mask_kernel(const double* __restrict__ xy, const int64_t* __restrict__ poly_off,
                                                         const int64_t* __restrict__ inst_poly_off, int H, int W,
                                                         uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned pm_bits[];      // (H*W + 1) toggle bits, then the edge prefix table
    const int inst = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const long nbits = (long)H * W + 1;
    const int nwords = (int)((nbits + 31) / 32);
    int* eoff = reinterpret_cast<int*>(pm_bits + nwords);                   // [k+1] prefix of points per edge (k <= 1023)
    uint8_t* o = out + (long)inst * H * W;
    const int p0 = (int)inst_poly_off[inst], p1 = (int)inst_poly_off[inst + 1];
    for (int q = p0; q < p1; ++q) {
        const double* pts = xy + 2 * poly_off[q];
        const int k = (int)(poly_off[q + 1] - poly_off[q]);
        for (int i = tid; i < nwords; i += nt) pm_bits[i] = 0u;
        if (tid == 0) {                                                     // points per edge: max(|dx|, |dy|) + 1
            int acc = 0;
            for (int e = 0; e < k; ++e) {
                const int e1 = (e + 1 == k) ? 0 : e + 1;
                const int xs = (int)(5.0 * pts[2 * e] + .5), ys = (int)(5.0 * pts[2 * e + 1] + .5);
                const int xe = (int)(5.0 * pts[2 * e1] + .5), ye = (int)(5.0 * pts[2 * e1 + 1] + .5);
                eoff[e] = acc;
                acc += max(abs(xe - xs), abs(ye - ys)) + 1;
            }
            eoff[k] = acc;
        }
        __syncthreads();
        const int M = k > 0 ? eoff[k] : 0;
        for (int j = tid + 1; j < M; j += nt) {                             // crossing test between walk points j-1 and j
            int lo = 0, hi = k;                                             // edge of point j: eoff[e] <= j < eoff[e+1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (eoff[mid] <= j) lo = mid; else hi = mid; }
            const PolyPt a = poly_edge_point(pts, k, lo, j - eoff[lo]);
            const int ep = (j - 1 >= eoff[lo]) ? lo : lo - 1;
            const PolyPt b = poly_edge_point(pts, k, ep, j - 1 - eoff[ep]);
            if (a.u == b.u) continue;
            double xd = (double)(a.u < b.u ? a.u : a.u - 1);
            xd = (xd + .5) / 5.0 - .5;
            if (floor(xd) != xd || xd < 0 || xd > W - 1) continue;
            double yd = (double)(a.v < b.v ? a.v : b.v);
            yd = (yd + .5) / 5.0 - .5;
            if (yd < 0) yd = 0; else if (yd > H) yd = H;
            yd = ceil(yd);
            const long pos = (long)(int)xd * H + (int)yd;
            atomicXor(&pm_bits[pos >> 5], 1u << (pos & 31));
        }
        __syncthreads();
        // running parity in column-major order.  carry into column x = parity of all toggles before x*H: per-column parities
        // first (the toggle words of a column are contiguous bits [x*H, (x+1)*H)), then an exclusive scan over the columns.
        int* cpar = eoff + 1024 + 1;                                        // [W] (W <= blockDim handled by the loops)
        for (int x = tid; x < W; x += nt) {
            unsigned par = 0u;
            const long b0 = (long)x * H, b1 = b0 + H;
            for (long wd = b0 >> 5; wd <= (b1 - 1) >> 5; ++wd) {
                unsigned w32 = pm_bits[wd];
                const long lo_b = wd << 5;
                if (lo_b < b0) w32 &= ~0u << (b0 - lo_b);
                if (lo_b + 32 > b1) w32 &= ~0u >> (lo_b + 32 - b1);
                par ^= (unsigned)__popc(w32);
            }
            cpar[x] = (int)(par & 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int x = 0; x < W; ++x) { const int c = cpar[x]; cpar[x] = run; run ^= c; }
        }
        __syncthreads();
        for (int x = tid; x < W; x += nt) {
            unsigned par = (unsigned)cpar[x];
            const long b0 = (long)x * H;
            for (int y = 0; y < H; ++y) {
                const long pos = b0 + y;
                par ^= (pm_bits[pos >> 5] >> (pos & 31)) & 1u;
                const uint8_t bit = (uint8_t)par;
                uint8_t* dst = o + (long)y * W + x;
                *dst = (q == p0) ? bit : (uint8_t)(*dst | bit);            // maskUtils.merge of an instance's polygons: union
            }
        }
        __syncthreads();
    }
    if (p1 == p0)
        for (long i = tid; i < (long)H * W; i += nt) o[i] = 0;
}

LOFT_EXPORT int loft_poly2mask(const double* xy, const int64_t* poly_offsets, const int64_t* inst_poly_offsets, int num_inst, int H,
                               int W, int max_vertices, uint8_t* out, void* stream) {
    if (num_inst <= 0) return 0;
    const long nbits = (long)H * W + 1;
    const size_t lds = (size_t)((nbits + 31) / 32) * 4 + (1024 + 1 + (size_t)W) * 4;
    if (max_vertices > 1023 || lds > 160 * 1024) return (int)hipErrorInvalidValue;
    hipFuncSetAttribute((const void*)poly2mask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(poly2mask_kernel, dim3(num_inst), dim3(1024), lds, (hipStream_t)stream, xy, poly_offsets, inst_poly_offsets, H, W,
                       out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- mask targets -------------------------------------------------------------------------------
// masks u8 [Ktot,H,W]; for RoI i: mask index gt_idx[i], box (already clipped to the image) boxes[i];
// out[i][S][S] = (RoIAlign_avg_aligned(mask, box, S, scale 1, adaptive grid) >= 0.5) as fp32 0/1.
__global__ __launch_bounds__(256) void mask_target_kernel(const uint8_t* __restrict__ masks, int H, int W,
                                                          const float* __restrict__ boxes, const int64_t* __restrict__ gt_idx, int S,
                                                          float* __restrict__ out, const int64_t* __restrict__ mask_addr) {
    const int i = blockIdx.x;
    const float4 r = reinterpret_cast<const float4*>(boxes)[i];
    // mask_addr (optional): device address of every instance mask, so per-image mask tensors need no concatenation
    const uint8_t* m = mask_addr ? reinterpret_cast<const uint8_t*>(mask_addr[gt_idx[i]]) : masks + (long)gt_idx[i] * H * W;
    const float start_w = r.x - 0.5f, start_h = r.y - 0.5f;
    const float rw = (r.z - 0.5f) - start_w, rh = (r.w - 0.5f) - start_h;
    const float bin_h = rh / (float)S, bin_w = rw / (float)S;
    const int grid_h = (int)ceilf(rh / (float)S), grid_w = (int)ceilf(rw / (float)S);
    const int cnt = grid_h * grid_w;
    const float count = (float)(cnt > 1 ? cnt : 1);
    for (int t = threadIdx.x; t < S * S; t += blockDim.x) {
        const int py = t / S, px = t - py * S;
        float acc = 0.f;
        for (int iy = 0; iy < grid_h; ++iy) {
            float y = start_h + py * bin_h + ((float)iy + .5f) * bin_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                float x = start_w + px * bin_w + ((float)ix + .5f) * bin_w / (float)grid_w;
                if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
                int y_low = (int)yy, x_low = (int)xx, y_high, x_high;
                if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else y_high = y_low + 1;
                if (x_low >= W - 1) { x_high = x_low = W - 1; xx = (float)x_low; } else x_high = x_low + 1;
                const float ly = yy - (float)y_low, lx = xx - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
                const float v = hy * hx * (float)m[y_low * W + x_low] + hy * lx * (float)m[y_low * W + x_high] +
                                ly * hx * (float)m[y_high * W + x_low] + ly * lx * (float)m[y_high * W + x_high];
                acc += v;
            }
        }
        out[(long)i * S * S + t] = (acc / count) >= 0.5f ? 1.f : 0.f;
    }
}
LOFT_EXPORT int loft_mask_target(const uint8_t* masks, int H, int W, const float* boxes, const int64_t* gt_idx, int64_t n,
                                 int S, float* out, const int64_t* mask_addr, void* stream) {
    if (n <= 0) return 0;
    if (!masks && !mask_addr) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mask_target_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, masks, H, W, boxes, gt_idx, S,
                       out, mask_addr);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- mask paste (inference) ---------------------------------------------------------------------
// FCNMaskHead.get_seg_masks -> _do_paste_mask (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:151-308):
// sigmoid of the SxS logits, bilinear grid_sample(align_corners=False, zero padding) of each instance into its box
// on the full image, threshold.  One thread per output pixel, only the pixels that can see the box are sampled.
__global__ void mask_paste_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, int N, int S, int img_h,
                                  int img_w, float thr, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, n = blockIdx.z;
    if (x >= img_w) return;
    const float4 b = reinterpret_cast<const float4*>(boxes)[n];
    // reference CPU path (skip_empty=True, one instance per chunk): only the tight integer region around the box
    if ((float)x < fmaxf(floorf(b.x) - 1.f, 0.f) || (float)x >= fminf(ceilf(b.z) + 1.f, (float)img_w) ||
        (float)y < fmaxf(floorf(b.y) - 1.f, 0.f) || (float)y >= fminf(ceilf(b.w) + 1.f, (float)img_h)) {
        out[((long)n * img_h + y) * img_w + x] = 0;
        return;
    }
    float gx = ((float)x + 0.5f - b.x) / (b.z - b.x) * 2.f - 1.f;
    float gy = ((float)y + 0.5f - b.y) / (b.w - b.y) * 2.f - 1.f;
    if (isinf(gx)) gx = 0.f;
    if (isinf(gy)) gy = 0.f;
    // grid_sample, align_corners=False: source coordinate = ((g + 1) * S - 1) / 2
    const float sx = ((gx + 1.f) * (float)S - 1.f) * 0.5f, sy = ((gy + 1.f) * (float)S - 1.f) * 0.5f;
    float v = 0.f;
    if (sx > -1.f && sx < (float)S && sy > -1.f && sy < (float)S) {
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float lx = sx - fx, ly = sy - fy;
        const float* m = logits + (long)n * S * S;
        auto at = [&](int yy, int xx) -> float {
            if (yy < 0 || yy >= S || xx < 0 || xx >= S) return 0.f;
            return 1.f / (1.f + expf(-m[yy * S + xx]));
        };
        v = at(y0, x0) * (1.f - ly) * (1.f - lx) + at(y0, x0 + 1) * (1.f - ly) * lx + at(y0 + 1, x0) * ly * (1.f - lx) +
            at(y0 + 1, x0 + 1) * ly * lx;
    }
    out[((long)n * img_h + y) * img_w + x] = v >= thr ? 1 : 0;
}
LOFT_EXPORT int loft_mask_paste(const float* logits, const float* boxes, int N, int S, int img_h, int img_w, float thr,
                                uint8_t* out, void* stream) {
    if (N <= 0) return 0;
    dim3 grid(loft_cdiv(img_w, 256), img_h, N);
    hipLaunchKernelGGL(mask_paste_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, boxes, N, S, img_h, img_w, thr, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- footprint = roof mask translated by the predicted offset (inference / evaluation) -----------------------------------------
// BONAI's offset is the footprint -> roof vector (tools/bonai/bonai_evaluation.py:41 offset_model='footprint2roof'; the annotation's
// 'offset' field, mmdet/datasets/bonai.py:184-196): footprint(y, x) = roof(y + oy, x + ox) with the offset rounded to whole pixels
// (round half away from zero, like numpy's / C's lround on the values the evaluation reads back), zero outside the image.  bstool
// translates the roof POLYGON by -offset and rasterises; on the bitmaps simple_test returns the translation is this shift.
// in / out uint8 [N, H, W]; offsets fp32 [N, 2] = (dx, dy).  16 output bytes per thread, byte-exact against the numpy shift oracle.
__global__ void mask_translate_kernel(const uint8_t* __restrict__ in, const float* __restrict__ offsets, int N, int H, int W,
                                      uint8_t* __restrict__ out) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (x0 >= W) return;
    const float fx = offsets[2 * n], fy = offsets[2 * n + 1];
    const int ox = (int)lroundf(fx), oy = (int)lroundf(fy);
    const int sy = y + oy;
    const uint8_t* src = in + ((long)n * H + sy) * W;
    uint8_t* dst = out + ((long)n * H + y) * W + x0;
    const bool row_ok = sy >= 0 && sy < H;
    if (row_ok && x0 + ox >= 0 && x0 + 15 + ox < W && x0 + 15 < W && ((W | (x0 + ox)) & 3) == 0 && (((size_t)dst) & 15) == 0) {
        // whole 16-byte run inside the row, source dword-aligned: four dword loads, one 16-byte store
        const unsigned* sp = reinterpret_cast<const unsigned*>(src + x0 + ox);
        uint4 v;
        v.x = sp[0]; v.y = sp[1]; v.z = sp[2]; v.w = sp[3];
        *reinterpret_cast<uint4*>(dst) = v;
        return;
    }
    for (int i = 0; i < 16 && x0 + i < W; ++i) {
        const int sx = x0 + i + ox;
        dst[i] = (row_ok && sx >= 0 && sx < W) ? src[sx] : (uint8_t)0;
    }
}
LOFT_EXPORT int loft_mask_translate(const uint8_t* masks, const float* offsets, int N, int H, int W, uint8_t* out, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    if (masks == out) return (int)hipErrorInvalidValue;
    dim3 grid(loft_cdiv(loft_cdiv(W, 16), 64), H, N);
    hipLaunchKernelGGL(mask_translate_kernel, grid, dim3(64), 0, (hipStream_t)stream, masks, offsets, N, H, W, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- RandomSampler on the device
// mmdet/core/bbox/samplers/random_sampler.py:31-75 + base_sampler.py:34-101 for a whole batch in one launch: per image, up to
// max_pos of the positives (gt_inds > 0) and then num - #sampled_pos of the negatives (gt_inds == 0), each a uniformly random
// subset (mode 1) or the first ones in index order (mode 0: the tests' injected sampling), written in ascending index order
// (base_sampler.py:86,96 `.unique()`), padded with index N-1 / valid 0.  Replaces rand + two top-k over all 261 888 anchors
// + two sorts + gathers (0.45 ms) by radix-select on hashed keys: one 1024-thread workgroup per image makes <= 6 passes over
// the image's gt_inds per class -- count, four 8-bit histogram passes that pin down the k-th smallest key exactly, and an
// ordered compaction (block scan) of the keys below it (ties: first in index order).  Integer path; deterministic for a seed.
__device__ __forceinline__ unsigned sample_key(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + (i + 1ull) * 0x9E3779B97F4A7C15ull;      // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}

__device__ __forceinline__ int block_sum_1024(int v, int* red) {   // every thread gets the total
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
}

// The bucket of a 256-bin histogram that holds rank krem (1-based): the first bin whose running count reaches it -> sel[0], and the
// rank inside it -> sel[1].  Scan over the first four waves; one thread walking the bins was 256 dependent LDS reads, eight times
// per launch (~80 of the RoI sampler's 84 us).  Called by every thread, between barriers.
__device__ __forceinline__ void sample_find_bucket(const int* hist, int krem, int* wsum, int* sel) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = tid < 256 ? hist[tid] : 0;
    int incl = h;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (wave < 4 && lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave < 4) {
        for (int w = 0; w < wave; ++w) incl += wsum[w];
        const int excl = incl - h;
        if (excl < krem && krem <= incl) { sel[0] = tid; sel[1] = krem - excl; }
    }
    __syncthreads();
}

// class code per box: 1 positive, 2 negative, 0 ignored -- 8x fewer bytes than the int64 gt_inds for the sampler's passes
__global__ void sample_codes_kernel(const int64_t* __restrict__ gt_inds, int N, int Np, uint8_t* __restrict__ code) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= Np) return;
    uint8_t c = 0;
    if (i < N) { const int64_t g = gt_inds[(size_t)b * N + i]; c = g > 0 ? 1 : (g == 0 ? 2 : 0); }
    code[(size_t)b * Np + i] = c;
}

#define SAMPLE_POOL 4096

// One 1024-thread workgroup per image; every pass reads the image's class codes 16 per lane (uint4), Np = N rounded up to 16.
__global__ __launch_bounds__(1024) void random_sample_kernel(const uint8_t* __restrict__ code, int N, int Np, int num, int max_pos,
                                                             int mode, unsigned long long seed, int P, int Q,
                                                             int64_t* __restrict__ pos_idx, uint8_t* __restrict__ pos_valid,
                                                             int64_t* __restrict__ neg_idx, uint8_t* __restrict__ neg_valid) {
    __shared__ int red[16];
    __shared__ int hist[256];
    __shared__ int wsum[16];
    __shared__ int sel[2];
    __shared__ int pool_idx[SAMPLE_POOL];
    __shared__ unsigned pool_key[SAMPLE_POOL];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4* cv = reinterpret_cast<const uint4*>(code + (size_t)b * Np);
    const int nvec = Np >> 4;
    const unsigned long long sd = seed ^ ((unsigned long long)(b + 1) * 0xD6E8FEB86659FD93ull);
    int taken_pos = 0;
    for (int cls = 0; cls < 2; ++cls) {
        int64_t* oidx = cls == 0 ? pos_idx + (size_t)b * P : neg_idx + (size_t)b * Q;
        uint8_t* oval = cls == 0 ? pos_valid + (size_t)b * P : neg_valid + (size_t)b * Q;
        const int cap = cls == 0 ? P : Q;
        const int limit = cls == 0 ? min(max_pos, P) : min(max(num - taken_pos, 0), Q);
        const unsigned want = cls == 0 ? 1u : 2u;
        const unsigned long long sdc = sd + cls;
        // ---- pass A: count the candidates and collect them, in index order, into the LDS pool (valid while cnt <= SAMPLE_POOL:
        // RPN positives, everything in the RoI sampler) -- the selection then never touches the image's codes again
        int cnt = 0;
        for (int v0 = 0; v0 < nvec; v0 += 1024) {
            const int v = v0 + tid;
            unsigned msk = 0u;
            if (v < nvec) {
                const uint4 q = cv[v];
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 16; ++e) msk |= (((w4[e >> 2] >> (8 * (e & 3))) & 255u) == want) ? (1u << e) : 0u;
            }
            const int val = __popc(msk);
            int incl = val;
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            __syncthreads();
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int woff = 0, tot = 0;
            for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
            int at = cnt + incl - val + woff;
            if (at < SAMPLE_POOL)
                while (msk) {
                    const int e = __builtin_ctz(msk);
                    msk &= msk - 1u;
                    if (at < SAMPLE_POOL) pool_idx[at] = v * 16 + e;
                    ++at;
                }
            cnt += tot;
        }
        __syncthreads();
        const int k = min(cnt, limit);
        // ---- threshold key T: the k-th smallest hashed key among the candidates (only when a strict subset is drawn at random)
        unsigned T = 0xffffffffu;
        int k_eq = 0;                                // how many candidates with key == T to take
        const bool thresh = mode == 1 && k < cnt && k > 0;
        bool pooled = false;
        int m = 0;
        if (cnt <= SAMPLE_POOL) {
            m = cnt;
            if (!thresh) {                            // everything, or the first k in index order: the head of the pool
                for (int j = tid; j < k; j += 1024) { oidx[j] = pool_idx[j]; oval[j] = 1; }
                for (int j = k + tid; j < cap; j += 1024) { oidx[j] = N - 1; oval[j] = 0; }
                if (cls == 0) taken_pos = k;
                __syncthreads();
                continue;
            }
            for (int j = tid; j < m; j += 1024) pool_key[j] = sample_key(sdc, (unsigned long long)pool_idx[j]);
            __syncthreads();
            pooled = true;
        } else if (thresh && cnt > 16 * k) {
            // ---- many candidates, few wanted (the RPN's 256 of ~250 000 negatives): ONE hashed pass keeps only the candidates
            // whose key is below a cut chosen for ~8k survivors, in index order; the exact selection then runs on that pool.
            const double cut = 8.0 * (double)k / (double)cnt * 4294967296.0;
            const unsigned T0 = cut >= 4294967295.0 ? 0xffffffffu : (unsigned)cut;
            for (int v0 = 0; v0 < nvec; v0 += 1024) {
                const int v = v0 + tid;
                unsigned msk = 0u;
                unsigned keys[16];
                if (v < nvec) {
                    const uint4 q = cv[v];
                    const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        keys[e] = 0u;
                        if (((w4[e >> 2] >> (8 * (e & 3))) & 255u) == want) {
                            keys[e] = sample_key(sdc, (unsigned long long)(v * 16 + e));
                            msk |= (keys[e] < T0) ? (1u << e) : 0u;
                        }
                    }
                }
                const int val = __popc(msk);
                int incl = val;
                for (int o = 1; o < 64; o <<= 1) {
                    const int u = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += u;
                }
                __syncthreads();
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                int woff = 0, tot = 0;
                for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
                int at = m + incl - val + woff;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((msk >> e) & 1u) {
                        if (at < SAMPLE_POOL) { pool_idx[at] = v * 16 + e; pool_key[at] = keys[e]; }
                        ++at;
                    }
                m += tot;
            }
            __syncthreads();
            pooled = m >= k && m <= SAMPLE_POOL;      // (else: the full passes below; probability ~1e-9 for the 8k cut)
        }
        if (pooled) {
            unsigned prefix = 0u;
            int krem = k;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
                for (int j = tid; j < m; j += 1024) {
                    const unsigned key = pool_key[j];
                    if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
                }
                __syncthreads();
                sample_find_bucket(hist, krem, wsum, sel);
                prefix |= (unsigned)sel[0] << shift;
                krem = sel[1];
                __syncthreads();
            }
            const unsigned Tp = prefix;
            const int keq = krem;
            int base_lt = 0, base_eq = 0;
            for (int j0 = 0; j0 < m; j0 += 1024) {      // the pool is in index order: the same ordered compaction, 1 entry per lane
                const int j = j0 + tid;
                int flt = 0, feq = 0;
                if (j < m) { const unsigned key = pool_key[j]; flt = key < Tp; feq = key == Tp; }
                const int val = flt | (feq << 16);
                int incl = val;
                for (int o = 1; o < 64; o <<= 1) {
                    const int u = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += u;
                }
                __syncthreads();
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                int woff = 0, tot = 0;
                for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
                const int excl = incl - val + woff;
                const int r_lt = base_lt + (excl & 0xffff), r_eq = base_eq + (excl >> 16);
                int pos = -1;
                if (flt) pos = r_lt + min(r_eq, keq);
                else if (feq && r_eq < keq) pos = r_lt + r_eq;
                if (pos >= 0 && pos < k) { oidx[pos] = pool_idx[j]; oval[pos] = 1; }
                base_lt += tot & 0xffff; base_eq += tot >> 16;
            }
            for (int j = k + tid; j < cap; j += 1024) { oidx[j] = N - 1; oval[j] = 0; }
            if (cls == 0) taken_pos = k;
            __syncthreads();
            continue;
        }
        if (thresh) {
            unsigned prefix = 0u;
            int krem = k;                            // rank (1-based) still to locate inside the current prefix bucket
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
                for (int v = tid; v < nvec; v += 1024) {
                    const uint4 q = cv[v];
                    const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (((w4[e >> 2] >> (8 * (e & 3))) & 255u) == want) {
                            const unsigned key = sample_key(sdc, (unsigned long long)(v * 16 + e));
                            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
                        }
                }
                __syncthreads();
                sample_find_bucket(hist, krem, wsum, sel);
                prefix |= (unsigned)sel[0] << shift;
                krem = sel[1];
                __syncthreads();
            }
            T = prefix;
            k_eq = krem;
        }
        // ---- ordered compaction: candidates with key < T plus the first k_eq with key == T (mode 0 / k == cnt: the first k)
        int base_lt = 0, base_eq = 0;                // running counts (block-uniform)
        for (int v0 = 0; v0 < nvec; v0 += 1024) {    // 16384 boxes per round, 16 consecutive ones per thread
            const int v = v0 + tid;
            unsigned mlt = 0u, meq = 0u;             // bit e: element e of this thread is below / at the threshold
            if (v < nvec) {
                const uint4 q = cv[v];
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (((w4[e >> 2] >> (8 * (e & 3))) & 255u) == want) {
                        if (thresh) {
                            const unsigned key = sample_key(sdc, (unsigned long long)(v * 16 + e));
                            mlt |= (key < T) ? (1u << e) : 0u;
                            meq |= (key == T) ? (1u << e) : 0u;
                        } else {
                            mlt |= 1u << e;          // first k in index order (everything when k == cnt)
                        }
                    }
            }
            const int val = __popc(mlt) | (__popc(meq) << 16);   // two 16-bit counters per scan word (<= 16384 per round)
            int incl = val;
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            __syncthreads();
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int woff = 0, tot = 0;
            for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
            const int excl = incl - val + woff;
            int r_lt = base_lt + (excl & 0xffff), r_eq = base_eq + (excl >> 16);
            unsigned m = mlt | meq;
            while (m) {
                const int e = __builtin_ctz(m);
                m &= m - 1u;
                if ((mlt >> e) & 1u) {
                    const int pos = r_lt + min(r_eq, k_eq);
                    if (pos < k) { oidx[pos] = (int64_t)v * 16 + e; oval[pos] = 1; }
                    ++r_lt;
                } else {
                    if (r_eq < k_eq) {
                        const int pos = r_lt + r_eq;
                        if (pos < k) { oidx[pos] = (int64_t)v * 16 + e; oval[pos] = 1; }
                    }
                    ++r_eq;
                }
            }
            base_lt += tot & 0xffff; base_eq += tot >> 16;
            if (base_lt + min(base_eq, k_eq) >= k) break;          // block-uniform early exit
        }
        for (int j = k + tid; j < cap; j += 1024) { oidx[j] = N - 1; oval[j] = 0; }
        if (cls == 0) taken_pos = k;
        __syncthreads();
    }
}
LOFT_EXPORT int64_t loft_random_sample_workspace_bytes(int B, int N) { return (int64_t)B * ((N + 15) / 16 * 16); }

LOFT_EXPORT int loft_random_sample(const int64_t* gt_inds, int B, int N, int num, int max_pos, int mode, uint64_t seed,
                                   int64_t* pos_idx, uint8_t* pos_valid, int64_t* neg_idx, uint8_t* neg_valid, void* workspace,
                                   void* stream) {
    if (B <= 0) return 0;
    if (N <= 0 || num < 0 || max_pos < 0 || (mode != 0 && mode != 1) || !workspace) return (int)hipErrorInvalidValue;
    const int P = max_pos < N ? max_pos : N, Q = num < N ? num : N;
    const int Np = (N + 15) / 16 * 16;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sample_codes_kernel, dim3(loft_cdiv(Np, 256), B), dim3(256), 0, s, gt_inds, N, Np, (uint8_t*)workspace);
    LOFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(random_sample_kernel, dim3(B), dim3(1024), 0, s, (const uint8_t*)workspace, N, Np, num, max_pos, mode,
                       (unsigned long long)seed, P, Q, pos_idx, pos_valid, neg_idx, neg_valid);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// The RPN losses' normaliser from the sampler's validity flags (anchor_head.py:363-364, 462-464):
//     avg = sum_b max(#pos_b, 1) + sum_b max(#neg_b, 1)
// One workgroup, one launch (it was two sum / clamp / sum chains, an add and a cast: eight launches of a few bytes each).
__global__ __launch_bounds__(256) void sampled_avg_factor_kernel(const uint8_t* __restrict__ pv, const uint8_t* __restrict__ nv, int B,
                                                                 int P, int Q, float* __restrict__ out) {
    __shared__ int part[4];
    int total = 0;
    for (int b = 0; b < B; ++b) {
        for (int side = 0; side < 2; ++side) {
            const uint8_t* v = side ? nv + (size_t)b * Q : pv + (size_t)b * P;
            const int n = side ? Q : P;
            int c = 0;
            for (int i = threadIdx.x; i < n; i += 256) c += v[i] != 0;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
            __syncthreads();
            const int cnt = part[0] + part[1] + part[2] + part[3];
            total += cnt > 1 ? cnt : 1;
        }
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}
LOFT_EXPORT int loft_sampled_avg_factor(const uint8_t* pos_valid, const uint8_t* neg_valid, int B, int P, int Q, float* out,
                                        void* stream) {
    if (B < 0 || P < 0 || Q < 0 || !out) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sampled_avg_factor_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pos_valid, neg_valid, B, P, Q, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- RPN: everything between the sampler and the losses
// For each sampled anchor (b, s) -- positives first, then negatives (anchor_head.py:187-237 `_get_targets_single`, :429-497
// `loss`): the anchor's pyramid level / pixel / slot, its objectness logit and 4 deltas gathered STRAIGHT from the fused head
// outputs (no [B, 261888, 5] flatten + concat), the classification label and weight, and for positives the regression target
// bbox2delta(anchor, assigned gt) (delta_xywh_bbox_coder.py:74-120).  One launch instead of ~45 small tensor ops.
struct RpnLevels {
    const float* head[8];
    int H[8], W[8];
    long off[9];
    int L;
};

__global__ void rpn_sample_gather_kernel(RpnLevels lv, int B, int Cp, int A, const float* __restrict__ anchors,
                                         const float* __restrict__ gts, int Kmax, const int64_t* __restrict__ gt_inds, long N,
                                         const int64_t* __restrict__ pidx, const uint8_t* __restrict__ pval, int P,
                                         const int64_t* __restrict__ nidx, const uint8_t* __restrict__ nval, int Q, Coder4 c,
                                         float* __restrict__ vals, int* __restrict__ rows, int64_t* __restrict__ slot_out,
                                         float* __restrict__ tgt, int64_t* __restrict__ label, float* __restrict__ weight) {
    const int S = P + Q;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const int b = (int)(i / S), s = (int)(i - (long)b * S);
    const bool is_pos = s < P;
    const long idx = is_pos ? pidx[(long)b * P + s] : nidx[(long)b * Q + (s - P)];
    const bool valid = is_pos ? pval[(long)b * P + s] != 0 : nval[(long)b * Q + (s - P)] != 0;
    int l = 0;
    for (int k = 1; k < lv.L; ++k) l = (idx >= lv.off[k]) ? k : l;
    const long local = idx - lv.off[l];
    const int pix = (int)(local / A), a = (int)(local - (long)pix * A);
    const int y = pix / lv.W[l], x = pix - y * lv.W[l];
    const float* hp = lv.head[l] + (((long)b * lv.H[l] + y) * lv.W[l] + x) * Cp;
    float* vo = vals + i * 5;
    vo[0] = hp[a];
#pragma unroll
    for (int j = 0; j < 4; ++j) vo[1 + j] = hp[A + 4 * a + j];
    reinterpret_cast<int4*>(rows)[i] = make_int4(b, valid ? l : -1, y, x);
    slot_out[i] = a;
    label[i] = (is_pos && valid) ? 1 : 0;
    weight[i] = valid ? 1.f : 0.f;
    if (is_pos) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            long gi = gt_inds[(long)b * N + idx] - 1;
            gi = gi < 0 ? 0 : gi;
            const float4 p = reinterpret_cast<const float4*>(anchors)[idx];
            const float4 g = reinterpret_cast<const float4*>(gts)[(long)b * Kmax + gi];
            const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f, pw = p.z - p.x, ph = p.w - p.y;
            const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
            o.x = ((gx - px) / pw - c.means[0]) / c.stds[0];
            o.y = ((gy - py) / ph - c.means[1]) / c.stds[1];
            o.z = (logf(gw / pw) - c.means[2]) / c.stds[2];
            o.w = (logf(gh / ph) - c.means[3]) / c.stds[3];
        }
        reinterpret_cast<float4*>(tgt)[(long)b * P + s] = o;
    }
}
LOFT_EXPORT int loft_rpn_sample_gather(const void* const* heads, const int* H, const int* W, const int64_t* lvl_off, int num_levels,
                                       int B, int Cp, int A, const float* anchors, const float* gts, int Kmax,
                                       const int64_t* gt_inds, int64_t N, const int64_t* pos_idx, const uint8_t* pos_valid, int P,
                                       const int64_t* neg_idx, const uint8_t* neg_valid, int Q, const float* means_host,
                                       const float* stds_host, float* vals, int32_t* rows, int64_t* slot, float* tgt,
                                       int64_t* label, float* weight, void* stream) {
    if (B <= 0 || P + Q <= 0) return 0;
    if (num_levels < 1 || num_levels > 8 || Kmax < 1) return (int)hipErrorInvalidValue;
    RpnLevels lv;
    for (int i = 0; i < num_levels; ++i) { lv.head[i] = (const float*)heads[i]; lv.H[i] = H[i]; lv.W[i] = W[i]; lv.off[i] = lvl_off[i]; }
    lv.off[num_levels] = lvl_off[num_levels];
    lv.L = num_levels;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    const long n = (long)B * (P + Q);
    hipLaunchKernelGGL(rpn_sample_gather_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, lv, B, Cp, A, anchors,
                       gts, Kmax, gt_inds, (long)N, pos_idx, pos_valid, P, neg_idx, neg_valid, Q, c, vals, rows, slot, tgt, label,
                       weight);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- RoI head: sampled RoIs, labels and regression targets
// SamplingResult + bbox2roi + BBoxHead.get_targets (sampling_result.py:25-53, transforms.py:54-73, bbox_head.py:84-138) for a
// batch in one launch.  The sampler writes its valid slots first, so entry (b, s) lands at off[b] + s (positives) or
// off[b] + npos[b] + (s - P) (negatives): per image [pos..., neg...] without a compaction scan; the counts come from one host
// read.  Positives also fill the mask / offset branches' lists (pos_rois, image index, assigned gt, row in the RoI list).
__global__ void roi_sample_targets_kernel(const float* __restrict__ cand, int Ncand, const int64_t* __restrict__ gt_inds,
                                          const float* __restrict__ gts, const int64_t* __restrict__ gt_labels, int Kmax,
                                          const int64_t* __restrict__ pidx, const int64_t* __restrict__ nidx, int P, int Q, int B,
                                          const int* __restrict__ npos, const int* __restrict__ nneg, const int* __restrict__ roff,
                                          const int* __restrict__ poff, int num_classes, Coder4 c, float* __restrict__ rois,
                                          int64_t* __restrict__ labels, float* __restrict__ label_w, float* __restrict__ tgt,
                                          float* __restrict__ tgt_w, float* __restrict__ pos_rois, int64_t* __restrict__ pos_b,
                                          int64_t* __restrict__ pos_gt, int64_t* __restrict__ pos_row) {
    const int S = P + Q;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const int b = (int)(i / S), s = (int)(i - (long)b * S);
    const bool is_pos = s < P;
    const int r = is_pos ? s : s - P;
    if (r >= (is_pos ? npos[b] : nneg[b])) return;
    const long idx = is_pos ? pidx[(long)b * P + r] : nidx[(long)b * Q + r];
    const int o = roff[b] + (is_pos ? r : npos[b] + r);
    const float4 box = reinterpret_cast<const float4*>(cand)[(long)b * Ncand + idx];
    float* ro = rois + (long)o * 5;
    ro[0] = (float)b; ro[1] = box.x; ro[2] = box.y; ro[3] = box.z; ro[4] = box.w;
    label_w[o] = 1.f;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    float tw = 0.f;
    if (is_pos) {
        long gi = gt_inds[(long)b * Ncand + idx] - 1;
        gi = gi < 0 ? 0 : gi;
        labels[o] = gt_labels[(long)b * Kmax + gi];
        const float4 g = reinterpret_cast<const float4*>(gts)[(long)b * Kmax + gi];
        const float px = (box.x + box.z) * 0.5f, py = (box.y + box.w) * 0.5f, pw = box.z - box.x, ph = box.w - box.y;
        const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
        t.x = ((gx - px) / pw - c.means[0]) / c.stds[0];
        t.y = ((gy - py) / ph - c.means[1]) / c.stds[1];
        t.z = (logf(gw / pw) - c.means[2]) / c.stds[2];
        t.w = (logf(gh / ph) - c.means[3]) / c.stds[3];
        tw = 1.f;
        const int po = poff[b] + r;
        float* pr = pos_rois + (long)po * 5;
        pr[0] = (float)b; pr[1] = box.x; pr[2] = box.y; pr[3] = box.z; pr[4] = box.w;
        pos_b[po] = b; pos_gt[po] = gi; pos_row[po] = o;
    } else {
        labels[o] = num_classes;
    }
    reinterpret_cast<float4*>(tgt)[o] = t;
    reinterpret_cast<float4*>(tgt_w)[o] = make_float4(tw, tw, tw, tw);
}
// The four tables of loft_roi_sample_targets from the sampler's validity flags, on the device: tab[0][b] = positives of image b,
// tab[1][b] = negatives, tab[2][b] = first row of image b in the RoI list, tab[3][b] = first row in the positives' list.  With
// this the host read of the counts is no longer in front of the launch: the caller sizes the outputs for the worst case, lets
// the first RoIAlign run on them, and fetches the counts meanwhile (bonai_amd.kernels.roi_sample_targets_begin).
__global__ __launch_bounds__(256) void roi_sample_offsets_kernel(const uint8_t* __restrict__ pval, const uint8_t* __restrict__ nval,
                                                                 int B, int P, int Q, int32_t* __restrict__ tab) {
    __shared__ int red[4];
    __shared__ int cnt[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = 0; b < B; ++b)
        for (int w = 0; w < 2; ++w) {
            const uint8_t* v = w == 0 ? pval + (long)b * P : nval + (long)b * Q;
            const int n = w == 0 ? P : Q;
            int c = 0;
            for (int i = tid; i < n; i += 256) c += v[i] ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            __syncthreads();
            if (lane == 0) red[wave] = c;
            __syncthreads();
            if (tid == 0) cnt[w][b] = red[0] + red[1] + red[2] + red[3];
        }
    __syncthreads();
    if (tid == 0) {
        int m = 0, np = 0;
        for (int b = 0; b < B; ++b) {
            tab[b] = cnt[0][b]; tab[B + b] = cnt[1][b]; tab[2 * B + b] = m; tab[3 * B + b] = np;
            m += cnt[0][b] + cnt[1][b]; np += cnt[0][b];
        }
    }
}
LOFT_EXPORT int loft_roi_sample_offsets(const uint8_t* pos_valid, const uint8_t* neg_valid, int B, int P, int Q, int32_t* tab,
                                        void* stream) {
    if (B <= 0) return 0;
    if (B > 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(roi_sample_offsets_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pos_valid, neg_valid, B, P, Q, tab);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_roi_sample_targets(const float* cand, int Ncand, const int64_t* gt_inds, const float* gts,
                                        const int64_t* gt_labels, int Kmax, const int64_t* pos_idx, const int64_t* neg_idx, int P,
                                        int Q, int B, const int32_t* npos_dev, const int32_t* nneg_dev, const int32_t* roi_off_dev,
                                        const int32_t* pos_off_dev, int num_classes, const float* means_host, const float* stds_host,
                                        float* rois, int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
                                        float* pos_rois, int64_t* pos_img, int64_t* pos_gt, int64_t* pos_row, void* stream) {
    if (B <= 0 || P + Q <= 0) return 0;
    Coder4 c;
    for (int i = 0; i < 4; ++i) { c.means[i] = means_host[i]; c.stds[i] = stds_host[i]; }
    const long n = (long)B * (P + Q);
    hipLaunchKernelGGL(roi_sample_targets_kernel, dim3(loft_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, cand, Ncand, gt_inds, gts,
                       gt_labels, Kmax, pos_idx, neg_idx, P, Q, B, npos_dev, nneg_dev, roi_off_dev, pos_off_dev, num_classes, c, rois,
                       labels, label_weights, bbox_targets, bbox_weights, pos_rois, pos_img, pos_gt, pos_row);
    LOFT_LAUNCH_CHECK();
    return 0;
}
