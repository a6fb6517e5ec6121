// nms.hip -- wavefront-parallel segmented greedy NMS + segmented score sort for gfx950.
//
// Replaces mmcv.ops.batched_nms -> nms  [mmcv-1.0.5] as called from
// mmdet/models/dense_heads/rpn_head.py:166-168 (per image, level ids as the batch index) and
// mmdet/core/post_processing/bbox_nms.py:63.  One launch pair handles EVERY (image, level)
// segment of the batch: batched_nms's coordinate shift makes segments mutually independent, so
// the N x N suppression matrix is block diagonal and is only ever built per segment.
//
//  (1) nms_mask_kernel : one 64-lane wavefront per 64x64 tile of the upper triangle; lane t owns
//      COLUMN box c*64+t and emits ONE 64-bit word (which row boxes of the tile suppress it) -- the
//      wave64 width IS the bitmask word width, so there is no cross-lane packing step.  Row boxes
//      are staged once in LDS.
//  (2) nms_scan_kernel : one wavefront per segment walks the 64-box chunks in order; the
//      in-chunk dependency chain is resolved in scalar registers with v_readlane (no memory) and
//      only for boxes that have a candidate suppressor inside their chunk; the kept set is then
//      applied to the later chunks' tiles lane-locally (column-form tiles: no cross-lane reduction).
//
// Integer / bit-exact path.  The suppression predicate is the division-free form of the
// mmcv-1.0.5 device kernel (inter > thr * union) and this file is compiled with
// -ffp-contract=off, so keep lists equal oracle/loft_oracle.c::orc_nms bit for bit.
// Total order for "sort by score descending": (score desc, original index asc) == stable radix
// sort, which is what loft_segmented_sort_desc provides (an in-house LSD radix sort, stable by
// construction).
#include "loft_common.h"
#include "../../include/loft_hip.h"

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

// Layout of the suppression matrix: 64 x 64 bit tiles, COLUMN form.  Tile (rt, ct) of a segment is 64 consecutive words; word t
// belongs to COLUMN box ct*64+t and bit j says "row box rt*64+j suppresses it".  Tile-row rt of segment s starts at word
// (nms_tile_row0(o_s, s) + rt) * max_words * 64: floor(o_s / 64) + s never overlaps the previous segment's last, partial tile-row.
// The scan then needs, per chunk of 64 row boxes, ONE coalesced 512-byte load per later chunk and (word & kept) != 0 per lane --
// no cross-lane reduction (the row form needed a 64-lane OR-reduce-scatter of 48 words per chunk through the LDS crossbar).
__device__ __forceinline__ long nms_tile_row0(long o, int seg) { return (o >> 6) + seg; }

// grid: (col_tile, row_tile, segment); block: 64 threads (one wave).
template <int PRED>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ seg_off,
                                                      const float* __restrict__ seg_shift, float thr, int max_words,
                                                      unsigned long long* __restrict__ mask, const float* __restrict__ img_max,
                                                      int levels) {
    const int seg = blockIdx.z, rt = blockIdx.y, ct = blockIdx.x;
    if (ct < rt) return;  // lower triangle never read
    const long o = seg_off[seg];
    const int n = (int)(seg_off[seg + 1] - o);
    if (rt * 64 >= n || ct * 64 >= n) return;
    // batched_nms's shift: given per segment, or derived here as level * (the image's largest coordinate + 1) -- segment s is level
    // s % levels of image s / levels (same two fp32 operations as the tensor expression; -ffp-contract=off)
    const float shift = seg_shift ? seg_shift[seg] : (img_max ? (float)(seg % levels) * (img_max[seg / levels] + 1.f) : 0.f);
    __shared__ float4 rb[64];
    const int t = threadIdx.x;
    const int ri = rt * 64 + t;
    if (ri < n) rb[t] = load_box(boxes, o + ri, shift);
    __syncthreads();
    const int cj = ct * 64 + t;
    unsigned long long m = 0ull;
    if (cj < n) {
        const float4 b = load_box(boxes, o + cj, shift);
        const int nrow = min(64, n - rt * 64);
        const int end = (rt == ct) ? min(t, nrow) : nrow;          // diagonal tile: only the rows above the column
        for (int j = 0; j < end; ++j)
            if (iou_gt<PRED>(rb[j], b, thr)) m |= (1ull << j);
    }
    mask[((size_t)(nms_tile_row0(o, seg) + rt) * max_words + ct) * 64 + t] = m;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffull), lane);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// grid: (segments); block: 64 threads (one wave).  Lane j stands for box c*64+j of the current chunk c AND keeps, as bit w of
// R[w >> 6], whether box w*64+j of chunk w is already suppressed.  Per chunk:
//   * rem_c = ballot(bit c of R); the in-chunk chain runs in scalar registers on the diagonal tile (column form: word j = the
//     earlier boxes of the chunk that suppress box j) and visits only boxes that HAVE such a candidate: a box nothing in its chunk
//     overlaps is kept iff it is alive.  (The first forms walked every kept box, ~100 cycles each for a lone wave: 2.7 us per
//     chunk, 241 us per launch on the proposal chain every RoI-head launch waits for.)
//   * for every later chunk w: R bit w |= (tile(c, w) word & kept) != 0 -- one coalesced load (requested a chunk ahead into the
//     other of two register sets; unconditional, so the compiler can count them in s_waitcnt) and six VALU instructions.
template <bool ONE>       // ONE: at most 64 chunks (4096 boxes) -- R is one word per lane, no slot select anywhere
__device__ __forceinline__ void nms_scan_body(const unsigned long long* __restrict__ mask, long o, int n, int seg, int max_words,
                                              uint8_t* __restrict__ keep) {
    const int lane = threadIdx.x;
    const int nwords = (n + 63) >> 6;
    constexpr int PF = 48;                 // tiles held in registers per chunk (segments up to 3136 boxes fully)
    constexpr int NR = ONE ? 1 : NMS_MAX_WORDS_PER_LANE;
    unsigned long long R[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) R[q] = 0ull;
    const unsigned long long* seg_tiles = mask + (size_t)nms_tile_row0(o, seg) * max_words * 64;

    // Plain loads off a wave-uniform tile-row pointer (scalar base + lane offset; no per-load predicate or select: a first version
    // with `need ? tile : base` per load compiled to ~20 instructions and an SGPR spill per tile).  Tiles past the segment's last
    // chunk are READ all the same -- inside the workspace (its size includes PF tiles of slack), unwritten or another segment's --
    // and only ever set R bits of chunks >= nwords, which nothing reads.
    auto fetch = [&](int c, unsigned long long (&T)[PF], unsigned long long& diag) {
        const int cc = min(c, nwords - 1);
        const unsigned long long* trow = seg_tiles + ((size_t)cc * max_words + cc) * 64;   // tile (cc, cc)
        diag = trow[lane];
#pragma unroll
        for (int k = 0; k < PF; ++k) T[k] = trow[(k + 1) * 64 + lane];
    };
    auto mark = [&](int w, unsigned long long t, unsigned long long kept) {
        const unsigned long long bit = (t & kept) != 0ull ? (1ull << (w & 63)) : 0ull;
        if constexpr (ONE) {
            R[0] |= bit;
        } else {
#pragma unroll
            for (int q = 0; q < NR; ++q) R[q] |= (q == (w >> 6)) ? bit : 0ull;
        }
    };
    auto process = [&](int c, unsigned long long (&T)[PF], unsigned long long diag) {
        const int row = c * 64 + lane;
        unsigned long long r = R[0];
        if constexpr (!ONE) {
#pragma unroll
            for (int q = 1; q < NR; ++q) r = (q == (c >> 6)) ? R[q] : r;
        }
        const unsigned long long rem_c = __ballot((r >> (c & 63)) & 1ull);
        const int nvalid = min(64, n - c * 64);
        const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        const unsigned long long alive = ~rem_c & valid;
        unsigned long long kept = alive;
        unsigned long long todo = __ballot(diag != 0ull) & alive;      // boxes with an in-chunk candidate suppressor
        while (todo) {  // wave-uniform scalar loop, ascending: every earlier box's fate is final when box b is looked at
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (readlane64(diag, b) & kept) kept &= ~(1ull << b);
        }
        if (row < n) keep[o + row] = (uint8_t)((kept >> lane) & 1ull);
        if (c + 1 >= nwords) return;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (c + 1 + k >= nwords) break;                            // wave-uniform
            mark(c + 1 + k, T[k], kept);
        }
        for (int w = c + 1 + PF; w < nwords; ++w)                      // segments beyond 3136 boxes: the remaining tiles, loaded here
            mark(w, seg_tiles[((size_t)c * max_words + w) * 64 + lane], kept);
    };
    unsigned long long T0[PF], T1[PF], d0, d1;
    fetch(0, T0, d0);
    for (int c = 0; c < nwords; c += 2) {
        fetch(c + 1, T1, d1);
        process(c, T0, d0);
        if (c + 1 >= nwords) break;
        fetch(c + 2, T0, d0);
        process(c + 1, T1, d1);
    }
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int64_t* __restrict__ seg_off, int max_words,
                                                      uint8_t* __restrict__ keep) {
    const int seg = blockIdx.x;
    const long o = seg_off[seg];
    const int n = (int)(seg_off[seg + 1] - o);
    if (n <= 0) return;
    if (n <= 4096) nms_scan_body<true>(mask, o, n, seg, max_words, keep);
    else nms_scan_body<false>(mask, o, n, seg, max_words, keep);
}

LOFT_EXPORT int64_t loft_nms_workspace_bytes(int64_t total_boxes, int64_t max_segment, int64_t num_segments) {
    int64_t words = (max_segment + 63) / 64;
    if (words < 1) words = 1;
    return (((total_boxes >> 6) + num_segments + 1) * words + 48) * 512;   // tile-rows (nms_tile_row0) x tiles per row x 64 words, + the scan's read-ahead
}

static int nms_launch(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev, const float* img_max_dev,
                      int levels, int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr, int predicate,
                      void* workspace, uint8_t* keep, void* stream) {
    if (predicate != LOFT_NMS_PRED_DEVICE && predicate != LOFT_NMS_PRED_CPU) return (int)hipErrorInvalidValue;
    if (num_segments <= 0 || total_boxes <= 0) return 0;
    const int max_words = (int)((max_segment + 63) / 64);          // tiles per tile-row
    if (max_words > 64 * NMS_MAX_WORDS_PER_LANE) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(max_words, max_words, num_segments);
    if (predicate == LOFT_NMS_PRED_DEVICE)
        hipLaunchKernelGGL(nms_mask_kernel<0>, grid, dim3(64), 0, s, boxes, seg_offsets_dev, seg_shift_dev, iou_thr, max_words,
                           (unsigned long long*)workspace, img_max_dev, levels);
    else
        hipLaunchKernelGGL(nms_mask_kernel<1>, grid, dim3(64), 0, s, boxes, seg_offsets_dev, seg_shift_dev, iou_thr, max_words,
                           (unsigned long long*)workspace, img_max_dev, levels);
    LOFT_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(num_segments), dim3(64), 0, s, (const unsigned long long*)workspace,
                       seg_offsets_dev, max_words, keep);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_nms_segmented_pred(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev,
                                        int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr,
                                        int predicate, void* workspace, uint8_t* keep, void* stream) {
    return nms_launch(boxes, seg_offsets_dev, seg_shift_dev, nullptr, 1, num_segments, total_boxes, max_segment, iou_thr, predicate,
                      workspace, keep, stream);
}

// segments = (image, level) pairs, image-major; the level shift is computed on the device from img_max_dev [num_segments / levels]
// (loft_rpn_decode_levels' per-image maximum coordinate): no host-visible shift table between the decode and the NMS
LOFT_EXPORT int loft_nms_segmented_levels(const float* boxes, const int64_t* seg_offsets_dev, const float* img_max_dev, int levels,
                                          int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr, int predicate,
                                          void* workspace, uint8_t* keep, void* stream) {
    if (levels < 1 || num_segments % levels || img_max_dev == nullptr) return (int)hipErrorInvalidValue;
    return nms_launch(boxes, seg_offsets_dev, nullptr, img_max_dev, levels, num_segments, total_boxes, max_segment, iou_thr, predicate,
                      workspace, keep, stream);
}

LOFT_EXPORT int loft_nms_segmented(const float* boxes, const int64_t* seg_offsets_dev, const float* seg_shift_dev,
                                   int num_segments, int64_t total_boxes, int64_t max_segment, float iou_thr,
                                   void* workspace, uint8_t* keep, void* stream) {
    return loft_nms_segmented_pred(boxes, seg_offsets_dev, seg_shift_dev, num_segments, total_boxes, max_segment, iou_thr,
                                   LOFT_NMS_PRED_DEVICE, workspace, keep, stream);
}

// ---------------------------------------------------------------- segmented stable sort, descending
// A *segmented* radix sort with one workgroup per segment -- 40 (image, level) segments of up to 196 608 keys -- left
// the chip idle (round 1, a library primitive: 2.3 ms per call).  Instead: ONE device-wide radix sort over 64-bit
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

// ---- the sort itself: an in-house stable LSD radix sort (round 5: hipCUB is gone from the library).  4 bits per pass over the
// 32 + ceil(log2(segments)) significant bits of the composite key.  A workgroup owns a TILE of 256 x 8 consecutive elements, a
// thread 8 CONSECUTIVE ones: (1) per-(digit, workgroup) counts, (2) one exclusive scan over the [16][workgroups] table in
// digit-major order, (3) every thread re-counts its run, the workgroup scans the 16 x 256 per-thread counts per digit, and each
// thread places its elements in order at base(digit, workgroup) + (same-digit elements of earlier threads) + (its own earlier
// ones) -- input order is preserved inside every digit bucket, which is what makes the whole sort stable.  Off the training
// step (both sorts of the step are loft_segmented_topk_desc); ~10 passes of three small launches.
namespace {
constexpr int RS_E = 8, RS_T = 256, RS_TILE = RS_E * RS_T;
}
__global__ __launch_bounds__(256) void rsort_count_kernel(const unsigned long long* __restrict__ keys, long n, int shift, int nb,
                                                          unsigned* __restrict__ counts) {
    __shared__ unsigned cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const long base = (long)blockIdx.x * RS_TILE + (long)threadIdx.x * RS_E;
    unsigned loc[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) loc[d] = 0u;
#pragma unroll
    for (int e = 0; e < RS_E; ++e) {
        if (base + e < n) {
            const int d = (int)((keys[base + e] >> shift) & 15ull);
#pragma unroll
            for (int q = 0; q < 16; ++q) loc[q] += (q == d) ? 1u : 0u;      // (static register indexing: no scratch)
        }
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        unsigned v = loc[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt[d], v);
    }
    __syncthreads();
    if (threadIdx.x < 16) counts[(long)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}
// exclusive scan of counts[0 .. m) in place, one workgroup of 1024 threads walking chunks of 1024
__global__ __launch_bounds__(1024) void rsort_scan_kernel(unsigned* __restrict__ counts, long m) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long c0 = 0; c0 < m; c0 += 1024) {
        const long i = c0 + threadIdx.x;
        const unsigned v = i < m ? counts[i] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < m) counts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void rsort_scatter_kernel(const unsigned long long* __restrict__ kin, const int32_t* __restrict__ vin,
                                                            unsigned long long* __restrict__ kout, int32_t* __restrict__ vout, long n,
                                                            int shift, int nb, const unsigned* __restrict__ bases) {
    __shared__ unsigned off[16 * RS_T];          // off[d * 256 + t]: next output position of thread t's digit-d elements
    __shared__ unsigned wtot[16][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long base = (long)blockIdx.x * RS_TILE + (long)t * RS_E;
    unsigned long long k[RS_E];
    unsigned loc[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) loc[d] = 0u;
#pragma unroll
    for (int e = 0; e < RS_E; ++e) {
        k[e] = base + e < n ? kin[base + e] : 0ull;
        if (base + e < n) {
            const int d = (int)((k[e] >> shift) & 15ull);
#pragma unroll
            for (int q = 0; q < 16; ++q) loc[q] += (q == d) ? 1u : 0u;
        }
    }
    // per digit: exclusive scan of the 256 per-thread counts (wave scan + the four wave totals)
    unsigned pre[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        unsigned inc = loc[d];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        pre[d] = inc - loc[d];
        if (lane == 63) wtot[d][wave] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        unsigned b = bases[(long)d * nb + blockIdx.x];
        for (int w = 0; w < wave; ++w) b += wtot[d][w];
        off[d * RS_T + t] = b + pre[d];
    }
    // (each thread touches only its own column of `off` from here on: no barrier needed)
#pragma unroll
    for (int e = 0; e < RS_E; ++e) {
        if (base + e < n) {
            const int d = (int)((k[e] >> shift) & 15ull);
            const unsigned pos = off[d * RS_T + t]++;
            kout[pos] = k[e];
            vout[pos] = vin[base + e];
        }
    }
}

LOFT_EXPORT int loft_segmented_sort_desc(const float* keys_in, float* keys_out, const int32_t* vals_in, int32_t* vals_out,
                                         int64_t num_items, int num_segments, const int64_t* seg_offsets_dev,
                                         void* workspace, int64_t* workspace_bytes, void* stream) {
    if (num_items > 0x7fffffffL || num_segments < 1) return (int)hipErrorInvalidValue;
    int seg_bits = 1;
    while ((1 << seg_bits) < num_segments) ++seg_bits;
    const int npass = (32 + seg_bits + 3) / 4;
    const int nb = loft_cdiv(num_items > 0 ? num_items : 1, RS_TILE);
    const size_t kbytes = ((size_t)num_items * 8 + 255) / 256 * 256, vbytes = ((size_t)num_items * 4 + 255) / 256 * 256;
    const size_t cbytes = ((size_t)16 * nb * 4 + 255) / 256 * 256;
    if (!workspace) {
        *workspace_bytes = (int64_t)(2 * kbytes + vbytes + cbytes);
        return 0;
    }
    if (num_items <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* kb[2] = {(unsigned long long*)workspace, (unsigned long long*)((char*)workspace + kbytes)};
    int32_t* vtmp = (int32_t*)((char*)workspace + 2 * kbytes);
    unsigned* counts = (unsigned*)((char*)workspace + 2 * kbytes + vbytes);
    hipLaunchKernelGGL(sort_build_keys_kernel, dim3(loft_cdiv(num_items, 256)), dim3(256), 0, s, keys_in, seg_offsets_dev,
                       num_segments, (long)num_items, kb[0]);
    LOFT_LAUNCH_CHECK();
    // values ping-pong between vals_out and vtmp so that the LAST pass writes vals_out
    int32_t* vb[2];
    vb[(npass - 1) & 1] = vtmp; vb[npass & 1] = vals_out;         // pass p writes vb[(p + 1) & 1]; the last (p = npass - 1) -> vb[npass & 1]
    const int32_t* vsrc = vals_in;
    for (int p = 0; p < npass; ++p) {
        hipLaunchKernelGGL(rsort_count_kernel, dim3(nb), dim3(256), 0, s, kb[p & 1], (long)num_items, 4 * p, nb, counts);
        hipLaunchKernelGGL(rsort_scan_kernel, dim3(1), dim3(1024), 0, s, counts, (long)16 * nb);
        hipLaunchKernelGGL(rsort_scatter_kernel, dim3(nb), dim3(256), 0, s, kb[p & 1], vsrc, kb[(p + 1) & 1], vb[(p + 1) & 1],
                           (long)num_items, 4 * p, nb, counts);
        LOFT_LAUNCH_CHECK();
        vsrc = vb[(p + 1) & 1];
    }
    hipLaunchKernelGGL(sort_extract_keys_kernel, dim3(loft_cdiv(num_items, 256)), dim3(256), 0, s, kb[npass & 1], (long)num_items, keys_out);
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
// mk (optional): a byte per key; keys whose byte is 0 count as -1.0 (the NMS keep flags: `where(keep, score, -1)` without a pass)
template <typename F>
__device__ __forceinline__ void topk_walk(const float* __restrict__ kp, const uint8_t* __restrict__ mk, int len, F&& f) {
    const int tid = threadIdx.x;
    if (mk != nullptr) {
        if (((reinterpret_cast<size_t>(kp) & 15) | (reinterpret_cast<size_t>(mk) & 3)) == 0) {
            const int n4 = len >> 2;
            for (int q = tid; q < n4; q += SNMS_THREADS) {
                const float4 a = reinterpret_cast<const float4*>(kp)[q];
                const unsigned m = reinterpret_cast<const unsigned*>(mk)[q];
                const int i = q * 4;
                f(topk_ord((m & 0xffu) ? a.x : -1.f), i); f(topk_ord((m & 0xff00u) ? a.y : -1.f), i + 1);
                f(topk_ord((m & 0xff0000u) ? a.z : -1.f), i + 2); f(topk_ord((m & 0xff000000u) ? a.w : -1.f), i + 3);
            }
            for (int i = (n4 << 2) + tid; i < len; i += SNMS_THREADS) f(topk_ord(mk[i] ? kp[i] : -1.f), i);
        } else {
            for (int i = tid; i < len; i += SNMS_THREADS) f(topk_ord(mk[i] ? kp[i] : -1.f), i);
        }
        return;
    }
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
                                                               const int64_t* __restrict__ out_off,
                                                               const uint8_t* __restrict__ key_mask) {
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
    const uint8_t* mk = key_mask ? key_mask + s0 : nullptr;
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
            topk_walk(kp, mk, len, [&](unsigned o, int) {
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
        topk_walk(kp, mk, len, [&](unsigned o, int i) { cand[i] = compose(o, i); });
    } else if (total_eq == need) {
        topk_walk(kp, mk, len, [&](unsigned o, int i) {      // one counter bump per wave, not per candidate (same-address LDS atomics)
            const bool take = o >= T;
            const unsigned long long tm = __ballot(take);
            if (tm != 0ull) {
                const int lane = (int)(threadIdx.x & 63);
                const int leader = __ffsll((long long)tm) - 1;
                unsigned base = 0u;
                if (lane == leader) base = atomicAdd(&s_cnt, (unsigned)__popcll(tm));
                base = (unsigned)__shfl((int)base, leader, 64);
                if (take) cand[base + (unsigned)__popcll(tm & ((1ull << lane) - 1ull))] = compose(o, i);
            }
        });
    } else {
        unsigned eq_base = 0u;
        for (int c0 = 0; c0 < len; c0 += SNMS_THREADS) {
            const int i = c0 + tid;
            const unsigned o = i < len ? topk_ord((mk == nullptr || mk[i]) ? kp[i] : -1.f) : 0u;
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
                                         const uint8_t* key_mask_dev, void* stream) {
    if (k < 1 || k > TOPK_MAX) return (int)hipErrorInvalidValue;
    if (num_segments <= 0) return 0;
    hipLaunchKernelGGL(seg_topk_kernel, dim3(num_segments), dim3(SNMS_THREADS), 0, (hipStream_t)stream, keys_in, vals_in,
                       seg_offsets_dev, k, keys_out, vals_out, out_offsets_dev, key_mask_dev);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- merge of sorted runs (the second stage of the two-stage top-k) ---------------------------------------------------------------
// Stage 1 left, per sub-range ("run"), its first <= k entries in stable descending order.  The first k of a segment's merged order:
// the rank of an entry is its position in its own run plus, for every other run of the segment, the number of entries that precede it
// -- keys >= its key in EARLIER runs (those hold lower original indices: ties go to them), keys > its key in later runs; a binary
// search per run, all runs of a thread's entry independent of each other.  Entries with rank < k are written at out0 + rank.  One
// thread per entry; ~10 searches of 12 steps against one workgroup streaming 30 000 candidates four times (88 us -> a few).
// run_off [nrun + 1]: run r = candidates [run_off[r], run_off[r + 1]); run_first[r] / run_count[r]: the runs of r's segment;
// run_out[r]: where the segment's head is written.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ ck, const int32_t* __restrict__ cv,
                                                         const int64_t* __restrict__ run_off, const int32_t* __restrict__ run_first,
                                                         const int32_t* __restrict__ run_count, const int64_t* __restrict__ run_out,
                                                         int k, float* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
    const int r = blockIdx.y;
    const long r0 = run_off[r];
    const int len = (int)(run_off[r + 1] - r0);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= len) return;
    const unsigned x = topk_ord(ck[r0 + p]);
    const int f = run_first[r], nr = run_count[r];
    int rank = p;
    // the searches of a group of MR runs advance in lock step: MR independent loads per step instead of a chain of runs x 12 round trips
    constexpr int MR = 12;
    for (int g = f; g < f + nr; g += MR) {
        long q0[MR];
        int lo[MR], hi[MR];
#pragma unroll
        for (int j = 0; j < MR; ++j) {
            const int q = g + j;
            const bool on = q < f + nr && q != r;
            q0[j] = on ? run_off[q] : 0l;
            lo[j] = 0;
            hi[j] = on ? (int)(run_off[q + 1] - q0[j]) : 0;
        }
        bool any = true;
        while (any) {
            any = false;
            unsigned y[MR];
#pragma unroll
            for (int j = 0; j < MR; ++j)               // (unconditional loads: a finished search re-reads the run's first key)
                y[j] = topk_ord(ck[q0[j] + (lo[j] < hi[j] ? (lo[j] + hi[j]) >> 1 : 0)]);
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                // run q is descending: its entries that precede x = the first position where "precedes" fails; ties go to EARLIER runs
                const int mid = (lo[j] + hi[j]) >> 1;
                const bool act = lo[j] < hi[j];
                const bool before = (g + j < r) ? (y[j] >= x) : (y[j] > x);
                lo[j] = (act && before) ? mid + 1 : lo[j];
                hi[j] = (act && !before) ? mid : hi[j];
                any |= lo[j] < hi[j];
            }
        }
#pragma unroll
        for (int j = 0; j < MR; ++j) rank += lo[j];
    }
    if (rank >= k) return;
    // (the canonical key image, -0.0 -> +0.0, as the one-stage kernel writes it)
    keys_out[run_out[r] + rank] = __uint_as_float((x & 0x80000000u) ? (x & 0x7fffffffu) : ~x);
    vals_out[run_out[r] + rank] = cv[r0 + p];
}
LOFT_EXPORT int loft_topk_merge_runs(const float* cand_keys, const int32_t* cand_vals, int num_runs, int max_run,
                                     const int64_t* run_offsets_dev, const int32_t* run_first_dev, const int32_t* run_count_dev,
                                     const int64_t* run_out_dev, int k, float* keys_out, int32_t* vals_out, void* stream) {
    if (k < 1 || num_runs < 0 || max_run < 0) return (int)hipErrorInvalidValue;
    if (num_runs == 0 || max_run == 0) return 0;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((max_run + 255) / 256, num_runs), dim3(256), 0, (hipStream_t)stream, cand_keys,
                       cand_vals, run_offsets_dev, run_first_dev, run_count_dev, run_out_dev, k, keys_out, vals_out);
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
