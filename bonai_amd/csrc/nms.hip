// nms.hip -- wavefront-parallel segmented greedy NMS + segmented score sort for gfx950.
//
// Replaces mmcv.ops.batched_nms -> nms  [mmcv-1.0.5] as called from
// mmdet/models/dense_heads/rpn_head.py:166-168 (per image, level ids as the batch index) and
// mmdet/core/post_processing/bbox_nms.py:63.  One launch pair handles EVERY (image, level)
// segment of the batch: batched_nms's coordinate shift makes segments mutually independent, so
// the N x N suppression matrix is block diagonal and is only ever built per segment.
//
//  (1) nms_mask_kernel : one 64-lane wavefront per 64x64 tile of the upper triangle; lane t owns
//      row box r*64+t and emits ONE 64-bit word -- the wave64 width IS the bitmask word width,
//      so there is no cross-lane packing step.  Column boxes are staged once in LDS.
//  (2) nms_scan_kernel : one wavefront per segment walks the 64-box chunks in order; the
//      in-chunk dependency chain is resolved in scalar registers with v_readlane (no memory),
//      then the rows of the kept boxes are OR-ed into the lane-distributed "removed" bitmap with
//      coalesced 8-byte-per-lane loads.
//
// Integer / bit-exact path.  The suppression predicate is the division-free form of the
// mmcv-1.0.5 device kernel (inter > thr * union) and this file is compiled with
// -ffp-contract=off, so keep lists equal oracle/loft_oracle.c::orc_nms bit for bit.
// Total order for "sort by score descending": (score desc, original index asc) == stable radix
// sort, which is what loft_segmented_sort_desc provides (hipCUB radix sort, stable by
// construction).
#include "loft_common.h"
#include "../../include/loft_hip.h"
#include <hipcub/hipcub.hpp>

#define NMS_MAX_WORDS_PER_LANE 8  // segments up to 64*64*8 = 32768 boxes

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
    float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = (a.z - a.x) * (a.w - a.y);
    float sb = (b.z - b.x) * (b.w - b.y);
    float uni = sa + sb - inter;
    return inter > thr * uni;
}

__device__ __forceinline__ float4 load_box(const float* boxes, long i, float shift) {
    float4 b = *reinterpret_cast<const float4*>(boxes + 4 * i);
    b.x += shift; b.y += shift; b.z += shift; b.w += shift;
    return b;
}

// grid: (col_tile, row_tile, segment); block: 64 threads (one wave).
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ seg_off,
                                                      const float* __restrict__ seg_shift, float thr, int max_words,
                                                      unsigned long long* __restrict__ mask) {
    const int seg = blockIdx.z, rt = blockIdx.y, ct = blockIdx.x;
    if (ct < rt) return;  // lower triangle never read
    const long o = seg_off[seg];
    const int n = (int)(seg_off[seg + 1] - o);
    if (rt * 64 >= n || ct * 64 >= n) return;
    const float shift = seg_shift ? seg_shift[seg] : 0.f;
    __shared__ float4 cb[64];
    const int t = threadIdx.x;
    const int cj = ct * 64 + t;
    if (cj < n) cb[t] = load_box(boxes, o + cj, shift);
    __syncthreads();
    const int ri = rt * 64 + t;
    if (ri >= n) return;
    const float4 a = load_box(boxes, o + ri, shift);
    const int ncol = min(64, n - ct * 64);
    unsigned long long m = 0ull;
    const int start = (rt == ct) ? t + 1 : 0;
    for (int j = start; j < ncol; ++j)
        if (iou_gt(a, cb[j], thr)) m |= (1ull << j);
    mask[(size_t)(o + ri) * max_words + ct] = m;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffull), lane);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// grid: (segments); block: 64 threads (one wave).
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int64_t* __restrict__ seg_off, int max_words,
                                                      uint8_t* __restrict__ keep) {
    const int seg = blockIdx.x;
    const long o = seg_off[seg];
    const int n = (int)(seg_off[seg + 1] - o);
    if (n <= 0) return;
    const int lane = threadIdx.x;
    const int nwords = (n + 63) >> 6;
    unsigned long long removed[NMS_MAX_WORDS_PER_LANE];
#pragma unroll
    for (int q = 0; q < NMS_MAX_WORDS_PER_LANE; ++q) removed[q] = 0ull;

    for (int c = 0; c < nwords; ++c) {
        const int row = c * 64 + lane;
        unsigned long long diag = 0ull;
        if (row < n) diag = mask[(size_t)(o + row) * max_words + c];
        // removed word of chunk c lives in lane (c & 63), slot (c >> 6)
        unsigned long long mine = 0ull;
#pragma unroll
        for (int q = 0; q < NMS_MAX_WORDS_PER_LANE; ++q)
            if (q == (c >> 6)) mine = removed[q];
        const unsigned long long rem_c = readlane64(mine, c & 63);
        const int nvalid = min(64, n - c * 64);
        const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        unsigned long long alive = ~rem_c & valid, kept = 0ull;
        while (alive) {  // wave-uniform scalar loop
            const int b = __builtin_ctzll(alive);
            kept |= (1ull << b);
            alive &= ~readlane64(diag, b);
            alive &= ~(1ull << b);
        }
        if (row < n) keep[o + row] = (uint8_t)((kept >> lane) & 1ull);
        // OR the rows of the kept boxes into the lane-distributed removed bitmap (words > c)
        unsigned long long kk = kept;
        while (kk) {
            const int b = __builtin_ctzll(kk);
            kk &= kk - 1ull;
            const unsigned long long* rp = mask + (size_t)(o + c * 64 + b) * max_words;
#pragma unroll
            for (int q = 0; q < NMS_MAX_WORDS_PER_LANE; ++q) {
                const int w = q * 64 + lane;
                if (w > c && w < nwords) removed[q] |= rp[w];
            }
        }
    }
}

LOFT_EXPORT int64_t loft_nms_workspace_bytes(int64_t total_boxes, int64_t max_segment) {
    int64_t words = (max_segment + 63) / 64;
    if (words < 1) words = 1;
    return total_boxes * words * 8;
}

LOFT_EXPORT int loft_nms_segmented(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev,
                                   int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr,
                                   void* workspace, uint8_t* keep, void* stream) {
    if (num_segments <= 0 || total_boxes <= 0) return 0;
    const int max_words = (int)((max_segment + 63) / 64);
    if (max_words > 64 * NMS_MAX_WORDS_PER_LANE) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(max_words, max_words, num_segments);
    hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(64), 0, s, boxes, seg_offsets_dev, seg_shift_dev, iou_thr, max_words,
                       (unsigned long long*)workspace);
    LOFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(num_segments), dim3(64), 0, s, (const unsigned long long*)workspace,
                       seg_offsets_dev, max_words, keep);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- segmented stable sort, descending
LOFT_EXPORT int loft_segmented_sort_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                         int64_t num_items, int num_segments, const int64_t* seg_offsets_dev,
                                         void* workspace, int64_t* workspace_bytes, void* stream) {
    size_t bytes = workspace ? (size_t)*workspace_bytes : 0;
    hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairsDescending(
        workspace, bytes, keys_in, keys_out, vals_in, vals_out, (int)num_items, num_segments, seg_offsets_dev,
        seg_offsets_dev + 1, 0, 32, (hipStream_t)stream);
    if (!workspace) *workspace_bytes = (int64_t)bytes;
    return (int)e;
}
