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
//      then the rows of the kept boxes (one row per lane, loaded ahead of the scalar loop) are
//      OR-reduced across the wave into the lane-distributed "removed" bitmap.
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

// PRED = LOFT_NMS_PRED_DEVICE (0): the division-free predicate of mmcv-1.0.5's CUDA kernel, inter > thr * union -- the form
// the reference's GPU training / tools/test.py runs execute, and the library default.  PRED = LOFT_NMS_PRED_CPU (1): the
// predicate of mmcv-1.0.5's host nms (nms_cpu), inter / union >= thr.  The two differ exactly AT the threshold (IoU == thr is
// kept by the device form, suppressed by the host form) and, rarely, by one rounding of the division next to it.
template <int PRED>
__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
    float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = (a.z - a.x) * (a.w - a.y);
    float sb = (b.z - b.x) * (b.w - b.y);
    float uni = sa + sb - inter;
    if constexpr (PRED == 0) return inter > thr * uni;
    else return inter / uni >= thr;
}

__device__ __forceinline__ float4 load_box(const float* boxes, long i, float shift) {
    float4 b = *reinterpret_cast<const float4*>(boxes + 4 * i);
    b.x += shift; b.y += shift; b.z += shift; b.w += shift;
    return b;
}

// grid: (col_tile, row_tile, segment); block: 64 threads (one wave).
template <int PRED>
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
        if (iou_gt<PRED>(a, cb[j], thr)) m |= (1ull << j);
    mask[(size_t)(o + ri) * max_words + ct] = m;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffull), lane);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl_xor64(unsigned long long v, int d) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xffffffffull), d, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), d, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// OR-reduce-scatter of 16 words per lane over the 64 lanes (recursive halving on lane bits 0..3, then two full exchanges):
// on return EVERY lane holds the OR over all lanes of word (lane & 15): 15 + 2 64-bit exchanges instead of 16 x 6.
__device__ __forceinline__ unsigned long long or_reduce_scatter16(const unsigned long long (&x)[16], int lane) {
    unsigned long long y[8], z[4], u[2];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {     // keep the words whose index bit 0 equals lane bit 0
        const unsigned long long mine = b0 ? x[2 * j + 1] : x[2 * j], send = b0 ? x[2 * j] : x[2 * j + 1];
        y[j] = mine | shfl_xor64(send, 1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long mine = b1 ? y[2 * j + 1] : y[2 * j], send = b1 ? y[2 * j] : y[2 * j + 1];
        z[j] = mine | shfl_xor64(send, 2);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned long long mine = b2 ? z[2 * j + 1] : z[2 * j], send = b2 ? z[2 * j] : z[2 * j + 1];
        u[j] = mine | shfl_xor64(send, 4);
    }
    unsigned long long v = (b3 ? u[1] : u[0]) | shfl_xor64(b3 ? u[0] : u[1], 8);
    v |= shfl_xor64(v, 16);
    v |= shfl_xor64(v, 32);
    return v;
}

// grid: (segments); block: 64 threads (one wave).  Per 64-box chunk c:
//   * lane t loads ITS row (box c*64+t): the diagonal word and, in 16-word (128-byte) batches, the words right of it -- all
//     loads are independent and issued BEFORE the chunk's scalar keep loop, so the loop hides their latency (the first form of
//     this kernel loaded one row per KEPT box inside a dependent loop: ~0.25 us of exposed latency per kept box, 0.87 ms for
//     the 3000-box RPN segments of a batch);
//   * the in-chunk chain is resolved with v_readlane on the diagonal word (no memory);
//   * rows of suppressed boxes are zeroed and each batch is OR-reduced across lanes with or_reduce_scatter16, which leaves
//     word w in the lanes with (lane & 15) == (w & 15): exactly where the lane-distributed "removed" bitmap keeps it
//     (word w in lane w & 63, slot w >> 6).
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int64_t* __restrict__ seg_off, int max_words,
                                                      uint8_t* __restrict__ keep) {
    const int seg = blockIdx.x;
    const long o = seg_off[seg];
    const int n = (int)(seg_off[seg + 1] - o);
    if (n <= 0) return;
    const int lane = threadIdx.x;
    const int nwords = (n + 63) >> 6;
    const int nbatch = (nwords + 15) >> 4;
    constexpr int PF = 3;                  // batches loaded ahead of the keep loop (covers segments up to 3072 boxes fully)
    unsigned long long removed[NMS_MAX_WORDS_PER_LANE];
#pragma unroll
    for (int q = 0; q < NMS_MAX_WORDS_PER_LANE; ++q) removed[q] = 0ull;

    auto load_batch = [&](const unsigned long long* rp, int wb, int c, unsigned long long (&x)[16]) {
        // words [wb*16, wb*16+16) of this lane's row; only words > c and < nwords were written by nms_mask_kernel
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int w = wb * 16 + k;
            x[k] = (rp != nullptr && w > c && w < nwords) ? rp[w] : 0ull;
        }
    };
    auto deposit = [&](unsigned long long v, int wb) {
        if ((lane >> 4) == (wb & 3)) {
#pragma unroll
            for (int q = 0; q < NMS_MAX_WORDS_PER_LANE; ++q)
                if (q == (wb >> 2)) removed[q] |= v;
        }
    };

    unsigned long long diag_next = lane < n ? mask[(size_t)(o + lane) * max_words] : 0ull;
    for (int c = 0; c < nwords; ++c) {
        const int row = c * 64 + lane;
        const unsigned long long* rp = row < n ? mask + (size_t)(o + row) * max_words : nullptr;
        const unsigned long long diag = diag_next;
        diag_next = (row + 64 < n) ? mask[(size_t)(o + row + 64) * max_words + c + 1] : 0ull;   // next chunk's diagonal word
        const int wb0 = (c + 1) >> 4;
        unsigned long long x[PF][16];
#pragma unroll
        for (int p = 0; p < PF; ++p) load_batch(wb0 + p < nbatch ? rp : nullptr, wb0 + p, c, x[p]);
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
        const bool i_am_kept = (kept >> lane) & 1ull;
        if (row < n) keep[o + row] = (uint8_t)i_am_kept;
        if (c + 1 >= nwords) break;
        // OR the rows of the kept boxes into the removed bitmap (words > c)
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (wb0 + p >= nbatch) break;                 // wave-uniform
            if (!i_am_kept) {
#pragma unroll
                for (int k = 0; k < 16; ++k) x[p][k] = 0ull;
            }
            deposit(or_reduce_scatter16(x[p], lane), wb0 + p);
        }
        for (int wb = wb0 + PF; wb < nbatch; ++wb) {      // segments beyond 3072 boxes: remaining batches, loaded here
            unsigned long long xx[16];
            load_batch(i_am_kept ? rp : nullptr, wb, c, xx);
            deposit(or_reduce_scatter16(xx, lane), wb);
        }
    }
}

LOFT_EXPORT int64_t loft_nms_workspace_bytes(int64_t total_boxes, int64_t max_segment) {
    int64_t words = (max_segment + 63) / 64;
    if (words < 1) words = 1;
    return total_boxes * words * 8;
}

LOFT_EXPORT int loft_nms_segmented_pred(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev,
                                        int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr,
                                        int predicate, void* workspace, uint8_t* keep, void* stream) {
    if (predicate != LOFT_NMS_PRED_DEVICE && predicate != LOFT_NMS_PRED_CPU) return (int)hipErrorInvalidValue;
    if (num_segments <= 0 || total_boxes <= 0) return 0;
    const int max_words = (int)((max_segment + 63) / 64);
    if (max_words > 64 * NMS_MAX_WORDS_PER_LANE) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(max_words, max_words, num_segments);
    if (predicate == LOFT_NMS_PRED_DEVICE)
        hipLaunchKernelGGL(nms_mask_kernel<0>, grid, dim3(64), 0, s, boxes, seg_offsets_dev, seg_shift_dev, iou_thr, max_words,
                           (unsigned long long*)workspace);
    else
        hipLaunchKernelGGL(nms_mask_kernel<1>, grid, dim3(64), 0, s, boxes, seg_offsets_dev, seg_shift_dev, iou_thr, max_words,
                           (unsigned long long*)workspace);
    LOFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(num_segments), dim3(64), 0, s, (const unsigned long long*)workspace,
                       seg_offsets_dev, max_words, keep);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_nms_segmented(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev,
                                   int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr,
                                   void* workspace, uint8_t* keep, void* stream) {
    return loft_nms_segmented_pred(boxes, seg_offsets_dev, seg_shift_dev, num_segments, total_boxes, max_segment, iou_thr,
                                   LOFT_NMS_PRED_DEVICE, workspace, keep, stream);
}

// ---------------------------------------------------------------- segmented stable sort, descending
// hipCUB's *segmented* radix sort gives one workgroup per segment -- with 40 (image, level) segments of up to
// 196 608 keys that left the chip idle (2.3 ms per call).  Instead: ONE device-wide radix sort over 64-bit
// composite keys  (segment id << 32) | ~orderable(score)  -- ascending order of the composite = segments in
// order, scores descending inside each; radix sort is stable, so equal scores keep input (index) order.
__global__ void sort_build_keys_kernel(const float* __restrict__ keys, const int64_t* __restrict__ seg_off, int nseg, long n,
                                       unsigned long long* __restrict__ comp) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = nseg;   // largest s with seg_off[s] <= i
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    unsigned u = __float_as_uint(keys[i]);
    if (u == 0x80000000u) u = 0u;                                     // -0.0 == +0.0 must tie (then index order decides)
    const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
    comp[i] = ((unsigned long long)lo << 32) | (unsigned)(~ord);
}
__global__ void sort_extract_keys_kernel(const unsigned long long* __restrict__ comp, long n, float* __restrict__ keys_out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned ord = ~(unsigned)(comp[i] & 0xffffffffull);
    const unsigned u = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
    keys_out[i] = __uint_as_float(u);
}

LOFT_EXPORT int loft_segmented_sort_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                         int64_t num_items, int num_segments, const int64_t* seg_offsets_dev,
                                         void* workspace, int64_t* workspace_bytes, void* stream) {
    int seg_bits = 1;
    while ((1 << seg_bits) < num_segments) ++seg_bits;
    const size_t kbytes = ((size_t)num_items * 8 + 255) / 256 * 256;
    size_t cub_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (const unsigned long long*)nullptr,
                                                     (unsigned long long*)nullptr, vals_in, vals_out, (int)num_items, 0,
                                                     32 + seg_bits, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (!workspace) {
        *workspace_bytes = (int64_t)(2 * kbytes + cub_bytes);
        return 0;
    }
    if (num_items <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* k0 = (unsigned long long*)workspace;
    unsigned long long* k1 = (unsigned long long*)((char*)workspace + kbytes);
    void* tmp = (char*)workspace + 2 * kbytes;
    hipLaunchKernelGGL(sort_build_keys_kernel, dim3(loft_cdiv(num_items, 256)), dim3(256), 0, s, keys_in, seg_offsets_dev,
                       num_segments, (long)num_items, k0);
    LOFT_LAUNCH_CHECK();
    e = hipcub::DeviceRadixSort::SortPairs(tmp, cub_bytes, k0, k1, vals_in, vals_out, (int)num_items, 0, 32 + seg_bits, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sort_extract_keys_kernel, dim3(loft_cdiv(num_items, 256)), dim3(256), 0, s, k1, (long)num_items, keys_out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- linear / naive / gaussian soft-NMS
// mmcv-1.0.5 ships soft_nms as a CPU-only op (the reference therefore round-trips every test image through the
// host: mmdet/core/post_processing/bbox_nms.py:63 with test_cfg.rcnn.nms.type='soft_nms').  This is the same
// in-place max-selection algorithm as oracle/loft_oracle.c::orc_soft_nms, one workgroup per call:
//   per output slot i:  (1) argmax over the live range [i, nb) with FIRST-position tie break (block reduction),
//                       (2) swap into slot i, (3) decay every live score against box i in parallel,
//                       (4) drop the boxes that fell below min_score with the SAME final arrangement the
//                           sequential "overwrite with the last element and re-examine" loop produces: holes left
//                           of the new end are filled, in ascending order, by the surviving tail elements taken in
//                           descending order (two block-wide scans instead of a serial loop).
// State lives in a caller-provided global workspace (L2 resident for the <= few thousand boxes of this path).
#define SNMS_THREADS 1024

struct SnmsState { float* x1; float* y1; float* x2; float* y2; float* sc; float* ar; int64_t* idx; };

__device__ __forceinline__ void snms_swap(const SnmsState& s, int a, int b) {
    float t;
    t = s.x1[a]; s.x1[a] = s.x1[b]; s.x1[b] = t;  t = s.y1[a]; s.y1[a] = s.y1[b]; s.y1[b] = t;
    t = s.x2[a]; s.x2[a] = s.x2[b]; s.x2[b] = t;  t = s.y2[a]; s.y2[a] = s.y2[b]; s.y2[b] = t;
    t = s.sc[a]; s.sc[a] = s.sc[b]; s.sc[b] = t;  t = s.ar[a]; s.ar[a] = s.ar[b]; s.ar[b] = t;
    int64_t u = s.idx[a]; s.idx[a] = s.idx[b]; s.idx[b] = u;
}

// exclusive block scan of one int per thread (1024 threads = 16 waves); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_exscan(int v, int* total, int* wsum /*[16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < SNMS_THREADS / 64; ++w) {
        const int t = wsum[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------- segmented top-k (in-house; round 4)
// What the training step needs from `scores.sort(descending=True)` + `[:nms_pre]` (rpn_head.py:129-136) and from the final
// `dets[:nms_post]` (rpn_head.py:166-168) is only the FIRST k entries of each segment's stable descending order -- 3000 of up to
// 196 608 anchors per (image, level), 1000-2000 of 12 768 candidates per image.  One workgroup per segment, no library:
//   1. radix SELECT of the k-th largest key: three histogram passes (11 + 11 + 10 bits of the order-preserving key image) in LDS;
//   2. one pass appends every key above that threshold -- and the threshold's own ties -- to an LDS candidate list (<= 4096 x 8 B);
//      ties that straddle position k are taken in INDEX order (an ordered chunked scan; only on that rare path), which is what a
//      stable sort followed by [:k] keeps;
//   3. bitonic sort of the candidates on (key descending, index ascending), written to the head of the segment's output range.
// Bit-identical to the first k entries of loft_segmented_sort_desc (tests/test_roi_nms_gpu.py); entries past k are not written.
// out_off (optional): segment s writes its head at out_off[s] instead of seg_off[s] -- the two-stage form for segments too long for
// one workgroup (kernels.segmented_topk_desc: top-k of every <= 20k-key sub-range into a compact candidate list, then top-k of the
// candidates back into the segment's own range; index order among equal keys survives both stages).
#define TOPK_MAX 4096
__device__ __forceinline__ unsigned topk_ord(float f) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;                                     // -0.0 == +0.0 must tie
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);                // larger value <-> larger unsigned
}
// Walk keys[0, len) with every thread of the block: f(order-preserving key image, index).  The body is 16-byte loads, four of
// them in flight per thread, when the segment starts on a 16-byte boundary -- one workgroup has to stream up to 768 KiB per pass, and
// with one 4-byte load per thread and loop trip each pass was a chain of ~190 dependent L2 / HBM round trips (217 us per launch).
template <typename F>
__device__ __forceinline__ void topk_walk(const float* __restrict__ kp, int len, F&& f) {
    const int tid = threadIdx.x;
    if ((reinterpret_cast<size_t>(kp) & 15) == 0) {
        const int n4 = len >> 2;
        const float4* k4 = reinterpret_cast<const float4*>(kp);
        int q = tid;
        for (; q + 3 * SNMS_THREADS < n4; q += 4 * SNMS_THREADS) {
            const float4 a = k4[q], b = k4[q + SNMS_THREADS], c = k4[q + 2 * SNMS_THREADS], d = k4[q + 3 * SNMS_THREADS];
            const float4 v[4] = {a, b, c, d};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (q + u * SNMS_THREADS) * 4;
                f(topk_ord(v[u].x), i); f(topk_ord(v[u].y), i + 1); f(topk_ord(v[u].z), i + 2); f(topk_ord(v[u].w), i + 3);
            }
        }
        for (; q < n4; q += SNMS_THREADS) {
            const float4 a = k4[q];
            const int i = q * 4;
            f(topk_ord(a.x), i); f(topk_ord(a.y), i + 1); f(topk_ord(a.z), i + 2); f(topk_ord(a.w), i + 3);
        }
        for (int i = (n4 << 2) + tid; i < len; i += SNMS_THREADS) f(topk_ord(kp[i]), i);
    } else {
        for (int i = tid; i < len; i += SNMS_THREADS) f(topk_ord(kp[i]), i);
    }
}

__global__ __launch_bounds__(SNMS_THREADS) void seg_topk_kernel(const float* __restrict__ keys, const int32_t* __restrict__ vals_in,
                                                               const int64_t* __restrict__ seg_off, int k,
                                                               float* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                               const int64_t* __restrict__ out_off) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long cand[TOPK_MAX];
    __shared__ int wsum[SNMS_THREADS / 64];
    __shared__ unsigned s_bin, s_need, s_cnt;
    const int tid = threadIdx.x;
    const long s0 = seg_off[blockIdx.x], s1 = seg_off[blockIdx.x + 1];
    const int len = (int)(s1 - s0);
    const int kk = k < len ? k : len;
    if (kk <= 0) return;
    const long o0 = out_off ? out_off[blockIdx.x] : s0;              // where this segment's head is written
    const float* kp = keys + s0;
    unsigned T = 0u, need = (unsigned)kk, total_eq = 0u;
    if (kk < len) {
        unsigned prefix = 0u, pmask = 0u;
        const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = shifts[pass], nb = 1 << nbits[pass];
            for (int b = tid; b < 2048; b += SNMS_THREADS) hist[b] = 0u;
            __syncthreads();
            // Scores cluster (sigmoid outputs share an exponent; suppressed candidates are all -1): in the first pass whole waves
            // hit ONE bin, and 64 same-address LDS atomics serialise.  When every counting lane of the wave has the same bin, one
            // lane adds the count.
            topk_walk(kp, len, [&](unsigned o, int) {
                const bool act = (o & pmask) == prefix;
                const unsigned bin = (o >> sh) & (unsigned)(nb - 1);
                const unsigned long long am = __ballot(act);
                if (am != 0ull) {
                    const int leader = __ffsll((long long)am) - 1;
                    const unsigned lb = (unsigned)__shfl((int)bin, leader, 64);
                    if (__ballot(act && bin == lb) == am) {
                        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (unsigned)__popcll(am));
                    } else if (act) {
                        atomicAdd(&hist[bin], 1u);
                    }
                }
            });
            __syncthreads();
            // bins in DESCENDING order, two per thread: the bin where the running count first reaches `need`
            int tot;
            const int b0 = nb - 1 - 2 * tid, b1 = nb - 2 - 2 * tid;
            const unsigned h0 = b0 >= 0 ? hist[b0] : 0u, h1 = b1 >= 0 ? hist[b1] : 0u;
            const unsigned before = (unsigned)block_exscan((int)(h0 + h1), &tot, wsum);
            if (before < need && need <= before + h0) { s_bin = (unsigned)b0; s_need = need - before; s_cnt = h0; }
            else if (before + h0 < need && need <= before + h0 + h1) { s_bin = (unsigned)b1; s_need = need - before - h0; s_cnt = h1; }
            __syncthreads();
            prefix |= s_bin << sh;
            pmask |= (unsigned)(nb - 1) << sh;
            need = s_need;
            total_eq = s_cnt;
            __syncthreads();
        }
        T = prefix;
    }
    // ---- candidates: everything above T in any order; T's ties in any order when all of them are taken, else by index
    const unsigned n_gt = (unsigned)kk - (kk < len ? need : 0u);
    auto compose = [](unsigned o, int i) { return ((unsigned long long)(~o) << 32) | (unsigned)i; };
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    if (kk == len) {
        topk_walk(kp, len, [&](unsigned o, int i) { cand[i] = compose(o, i); });
    } else if (total_eq == need) {
        topk_walk(kp, len, [&](unsigned o, int i) {
            if (o >= T) cand[atomicAdd(&s_cnt, 1u)] = compose(o, i);
        });
    } else {
        unsigned eq_base = 0u;
        for (int c0 = 0; c0 < len; c0 += SNMS_THREADS) {
            const int i = c0 + tid;
            const unsigned o = i < len ? topk_ord(kp[i]) : 0u;
            const bool in = i < len;
            if (in && o > T) cand[atomicAdd(&s_cnt, 1u)] = compose(o, i);
            const int fe = (in && o == T) ? 1 : 0;
            int te;
            const unsigned pe = (unsigned)block_exscan(fe, &te, wsum);
            if (fe && eq_base + pe < need) cand[n_gt + eq_base + pe] = compose(o, i);
            eq_base += (unsigned)te;
        }
    }
    int P = 1;
    while (P < kk) P <<= 1;
    __syncthreads();
    for (int i = kk + tid; i < P; i += SNMS_THREADS) cand[i] = ~0ull;
    __syncthreads();
    // compare-exchange t of a sub-pass with distance j <= 64 touches elements [128 * (t / 64), +128) only: a wave's 64 exchanges stay
    // inside its own 128 elements for every such j, so those sub-passes need no workgroup barrier (a wave's LDS operations complete
    // in order) -- 20 barriers instead of 78 for 4096 candidates
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += SNMS_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long a = cand[lo], b = cand[hi];
                const bool up = (lo & k2) == 0;
                if ((a > b) == up) { cand[lo] = b; cand[hi] = a; }
            }
            if (j > 64 || j == 1) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    }
    for (int r = tid; r < kk; r += SNMS_THREADS) {
        const unsigned long long c = cand[r];
        const int i = (int)(unsigned)(c & 0xffffffffull);
        const unsigned o = ~(unsigned)(c >> 32);
        const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
        keys_out[o0 + r] = __uint_as_float(u);
        vals_out[o0 + r] = vals_in ? vals_in[s0 + i] : (int32_t)(s0 + i);
    }
}
LOFT_EXPORT int loft_segmented_topk_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                         int num_segments, const int64_t* seg_offsets_dev, int k, const int64_t* out_offsets_dev,
                                         void* stream) {
    if (k < 1 || k > TOPK_MAX) return (int)hipErrorInvalidValue;
    if (num_segments <= 0) return 0;
    hipLaunchKernelGGL(seg_topk_kernel, dim3(num_segments), dim3(SNMS_THREADS), 0, (hipStream_t)stream, keys_in, vals_in,
                       seg_offsets_dev, k, keys_out, vals_out, out_offsets_dev);
    LOFT_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(SNMS_THREADS) void soft_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                               int n, float iou_thr, float sigma, float min_score, int method,
                                                               SnmsState s, float* __restrict__ dets, int64_t* __restrict__ inds,
                                                               int* __restrict__ n_out) {
    __shared__ float red_v[SNMS_THREADS / 64];
    __shared__ int red_p[SNMS_THREADS / 64];
    __shared__ int wsum[SNMS_THREADS / 64];
    __shared__ int s_maxpos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = tid; p < n; p += SNMS_THREADS) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[p];
        s.x1[p] = b.x; s.y1[p] = b.y; s.x2[p] = b.z; s.y2[p] = b.w;
        s.sc[p] = scores[p]; s.ar[p] = (b.z - b.x) * (b.w - b.y); s.idx[p] = p;
    }
    __syncthreads();
    int nb = n;
    const int per = (n + SNMS_THREADS - 1) / SNMS_THREADS;   // contiguous slab of positions per thread (keeps scans ordered)
    for (int i = 0; i < nb; ++i) {
        // (1) first-position argmax over [i, nb)
        float bv = -3.0e38f;
        int bp = 0x7fffffff;
        for (int p = i + tid; p < nb; p += SNMS_THREADS) {
            const float v = s.sc[p];
            if (v > bv) { bv = v; bp = p; }      // ascending p per thread -> first maximum kept
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int op = __shfl_xor(bp, o, 64);
            if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
        }
        if (lane == 0) { red_v[wave] = bv; red_p[wave] = bp; }
        __syncthreads();
        if (tid == 0) {
            float v = red_v[0]; int p = red_p[0];
            for (int w = 1; w < SNMS_THREADS / 64; ++w)
                if (red_v[w] > v || (red_v[w] == v && red_p[w] < p)) { v = red_v[w]; p = red_p[w]; }
            s_maxpos = p;
            // (2) swap the winner into slot i and emit it
            snms_swap(s, i, p);
            dets[5 * i] = s.x1[i]; dets[5 * i + 1] = s.y1[i]; dets[5 * i + 2] = s.x2[i]; dets[5 * i + 3] = s.y2[i];
            dets[5 * i + 4] = s.sc[i];
            inds[i] = s.idx[i];
        }
        __syncthreads();
        const float ix1 = s.x1[i], iy1 = s.y1[i], ix2 = s.x2[i], iy2 = s.y2[i], iar = s.ar[i];
        // (3) decay; (4) survivor bookkeeping on contiguous slabs [lo, hi) of the live tail (i, nb)
        const int lo = i + 1 + tid * per, hi = min(nb, lo + per);
        int surv = 0;
        for (int p = lo; p < hi; ++p) {
            const float xx1 = fmaxf(ix1, s.x1[p]), yy1 = fmaxf(iy1, s.y1[p]);
            const float xx2 = fminf(ix2, s.x2[p]), yy2 = fminf(iy2, s.y2[p]);
            const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (iar + s.ar[p] - inter);
            float weight = 1.f;
            if (method == 0) { if (ovr >= iou_thr) weight = 0.f; }
            else if (method == 1) { if (ovr >= iou_thr) weight = 1.f - ovr; }
            else { weight = expf(-(ovr * ovr) / sigma); }
            const float v = s.sc[p] * weight;
            s.sc[p] = v;
            surv += (v >= min_score) ? 1 : 0;    // the sequential loop drops sc < min_score
        }
        int total_surv;
        const int surv_before = block_exscan(surv, &total_surv, wsum);
        const int nb_new = i + 1 + total_surv;
        if (nb_new < nb) {
            // holes: dead positions < nb_new, ranked ascending; fillers: survivors at positions >= nb_new, ranked descending
            int holes = 0, fill = 0;
            for (int p = lo; p < hi; ++p) {
                const bool alive = s.sc[p] >= min_score;
                holes += (!alive && p < nb_new) ? 1 : 0;
                fill += (alive && p >= nb_new) ? 1 : 0;
            }
            int tot_h, tot_f;
            const int hole_rank0 = block_exscan(holes, &tot_h, wsum);
            const int fill_before = block_exscan(fill, &tot_f, wsum);   // ascending rank; descending rank = tot_f-1-asc
            // publish filler positions by descending rank into the (now free) dets tail as scratch: use inds tail
            int r = fill_before;
            for (int p = lo; p < hi; ++p)
                if (s.sc[p] >= min_score && p >= nb_new) { inds[n - 1 - (tot_f - 1 - r)] = p; ++r; }  // slot n-1-d holds rank d
            __syncthreads();
            int hr = hole_rank0;
            for (int p = lo; p < hi; ++p)
                if (!(s.sc[p] >= min_score) && p < nb_new) {
                    const int src = (int)inds[n - 1 - hr];
                    s.x1[p] = s.x1[src]; s.y1[p] = s.y1[src]; s.x2[p] = s.x2[src]; s.y2[p] = s.y2[src];
                    s.sc[p] = s.sc[src]; s.ar[p] = s.ar[src]; s.idx[p] = s.idx[src];
                    ++hr;
                }
            (void)surv_before;
            nb = nb_new;
        }
        __syncthreads();
    }
    if (tid == 0) *n_out = nb;
}

LOFT_EXPORT int64_t loft_soft_nms_workspace_bytes(int64_t n) { return n * (6 * 4 + 8) + 64; }

LOFT_EXPORT int loft_soft_nms(const float* boxes, const float* scores, int64_t n, float iou_thr, float sigma, float min_score,
                              int method, void* workspace, float* dets, int64_t* inds, int* n_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) return (int)hipMemsetAsync(n_out, 0, sizeof(int), st);
    char* w = (char*)workspace;
    SnmsState s;
    s.idx = (int64_t*)w; w += 8 * n;
    s.x1 = (float*)w; w += 4 * n; s.y1 = (float*)w; w += 4 * n; s.x2 = (float*)w; w += 4 * n; s.y2 = (float*)w; w += 4 * n;
    s.sc = (float*)w; w += 4 * n; s.ar = (float*)w;
    hipLaunchKernelGGL(soft_nms_kernel, dim3(1), dim3(SNMS_THREADS), 0, st, boxes, scores, (int)n, iou_thr, sigma, min_score,
                       method, s, dets, inds, n_out);
    LOFT_LAUNCH_CHECK();
    return 0;
}
