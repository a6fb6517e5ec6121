// roi_align.hip -- multi-level fused RoIAlign (fwd + bwd) for gfx950.
//
// Replaces, in ONE launch per extractor, the reference's per-level gather -> mmcv.ops.RoIAlign ->
// scatter loop (mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:53-80) and the
// level rule (:32-51).  Arithmetic follows the mmcv-1.0.5 roi_align op (aligned=True, avg pool,
// sampling_ratio=0 -> adaptive ceil(roi/P) grid), see oracle/loft_oracle.c for the restatement.
//
// Layout (MI355X-first): feature maps are NHWC so the 256 channels of one pixel are one contiguous
// 512 B (bf16) / 1 KiB (fp32) run; a 64-lane wavefront owns one output bin and each lane reads 4
// consecutive channels of every bilinear corner -> every corner read is one fully coalesced wave
// access.  Output is [K, P, P, C] (NHWC over RoIs) in the feature dtype, optionally written as the
// four FOA rotations [4, K, P, P, C] (offset_head_expand_feature.py:163-196 == rot90 permutation).
// HBM-bound: algorithmic bytes per RoI = C*(Fh*Fw + P*P)*sizeof(T)   (SURVEY.md section 8d).
//
// Compiled with -ffp-contract=off so sample weights round exactly like the C oracle.
#include "loft_common.h"
#include <type_traits>
#include "../../include/loft_hip.h"

struct RoiLevels {
    const void* feat[4];
    int H[4], W[4];
    float scale[4];
    int num_levels;
    int finest_scale;
};

struct Bil {
    int y_low, x_low, y_high, x_high;
    float w1, w2, w3, w4;
    bool valid;
};

__device__ __forceinline__ Bil bil_setup(float y, float x, int height, int width) {
    Bil b;
    b.valid = !(y < -1.0f || y > (float)height || x < -1.0f || x > (float)width);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= width - 1)  { x_high = x_low = width - 1;  x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    b.y_low = y_low; b.x_low = x_low; b.y_high = y_high; b.x_high = x_high;
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
    return b;
}

struct RoiGeom {
    float start_h, start_w, bin_h, bin_w;
    int grid_h, grid_w, batch, level;
    float count;
};

__device__ __forceinline__ int roi_level(const float* roi, int num_levels, int finest_scale) {
    // single_level_roi_extractor.py:47-50
    float scale = sqrtf((roi[3] - roi[1]) * (roi[4] - roi[2]));
    float lv = floorf(log2f(scale / (float)finest_scale + 1e-6f));
    lv = fminf(fmaxf(lv, 0.f), (float)(num_levels - 1));
    return (int)lv;
}

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, const RoiLevels& L, int P) {
    RoiGeom g;
    g.batch = (int)roi[0];
    g.level = L.num_levels > 1 ? roi_level(roi, L.num_levels, L.finest_scale) : 0;
    float s = L.scale[g.level];
    g.start_w = roi[1] * s - 0.5f;
    g.start_h = roi[2] * s - 0.5f;
    float end_w = roi[3] * s - 0.5f, end_h = roi[4] * s - 0.5f;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    g.bin_h = rh / (float)P;
    g.bin_w = rw / (float)P;
    g.grid_h = (int)ceilf(rh / (float)P);
    g.grid_w = (int)ceilf(rw / (float)P);
    int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

// FOA rotation k of bin (py,px) on a PxP map: position of in[py][px] inside rot90(in, k).
__device__ __forceinline__ int rot_pos(int py, int px, int P, int k) {
    int i, j;
    switch (k) {
        case 0: i = py; j = px; break;
        case 1: i = P - 1 - px; j = py; break;
        case 2: i = P - 1 - py; j = P - 1 - px; break;
        default: i = px; j = P - 1 - py; break;
    }
    return i * P + j;
}

// Which RoI a workgroup of the forward kernels serves.  Without a launch order: RoI blockIdx.x.  With one (loft_roi_order: the
// list bucketed by (image, level, row strip of the level's map)): workgroups are dispatched round-robin over the 8 XCDs, each
// with its own 4 MiB L2, so workgroup b takes entry (b % 8) * ceil(K / 8) + b / 8 -- every XCD walks ONE contiguous eighth of
// the ordered list and the windows of RoIs that overlap meet in that XCD's L2 instead of being fetched from HBM once per RoI
// (sampling order: 1.4 GB of window reads for the 8192-RoI bbox list against 0.36 GB of maps).  Grid = 8 * ceil(K / 8).
__device__ __forceinline__ int roi_of_block(const int32_t* __restrict__ order, int K, int b = blockIdx.x) {
    if (!order) return b < K ? b : -1;
    const int per = (K + 7) >> 3, j = (b & 7) * per + (b >> 3);
    return j < K ? order[j] : -1;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(RoiLevels L, const float* __restrict__ rois, int K, int C,
                                                            int P, int n_rot, T* __restrict__ out,
                                                            const int32_t* __restrict__ order) {
    const int k = roi_of_block(order, K);
    if (k < 0) return;
    const float* roi = rois + 5 * (size_t)k;
    const RoiGeom g = roi_geom(roi, L, P);
    const int H = L.H[g.level], W = L.W[g.level];
    const T* fb = reinterpret_cast<const T*>(L.feat[g.level]) + (size_t)g.batch * H * W * C;
    const int cg = C >> 2;
    const int total = P * P * cg;
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        const int bin = w / cg, c0 = (w - bin * cg) << 2;
        const int py = bin / P, px = bin - py * P;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                const Bil b = bil_setup(y, x, H, W);
                if (!b.valid) continue;
                float v1[4], v2[4], v3[4], v4[4];
                ld4(fb + ((size_t)b.y_low * W + b.x_low) * C + c0, v1);
                ld4(fb + ((size_t)b.y_low * W + b.x_high) * C + c0, v2);
                ld4(fb + ((size_t)b.y_high * W + b.x_low) * C + c0, v3);
                ld4(fb + ((size_t)b.y_high * W + b.x_high) * C + c0, v4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += b.w1 * v1[q] + b.w2 * v2[q] + b.w3 * v3[q] + b.w4 * v4[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] /= g.count;
        for (int r = 0; r < n_rot; ++r) {
            const int pos = rot_pos(py, px, P, r);
            st4(out + (((size_t)r * K + k) * P * P + pos) * C + c0, acc);
        }
    }
}

// ---- forward, separable form (bf16 training path) -----------------------------------------------------------
// The adaptive-grid average of bilinear samples factorises per axis:
//     out[py,px,:] = 1/count * sum_y sum_x WY[py][y] * WX[px][x] * f[y,x,:]
// with WY[py][y] = sum over the bin's sample rows of the (clamped) hat weight of pixel row y (axis_weight, the table
// the backward kernel uses).  The sample-by-sample kernel above reads 4 corners per sample: P*P*grid^2*4 coalesced row
// reads per RoI (3136 for a 28x28-pixel footprint at P = 7); here wave w owns bin rows py = w, w+4, .. and walks the
// footprint columns once per bin row: colsum(x) = sum_{y in rows(py)} WY[py][y] * f[y,x,:] (rows(py) ~ bin_h + 2 reads),
// then adds colsum(x) * WX[p..p+2][x] to a sliding window of three bin accumulators (a pixel column gets weight from at
// most 3 bins when bin_w >= 1) -- (Fh + 2P) * Fw reads instead, 2-2.6x fewer, all register-static.  RoIs with sub-pixel
// bins or footprints beyond RF_MAX pixels take the sample loop.  Same clamping / validity rules; the fp32 sum order
// differs from the oracle's, so this form serves bf16 only (parity tolerance 8e-3 of the map's range).
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int pix, int size);
#define RF_MAX 64
#define RF_MAXP 14

__device__ __forceinline__ void roi_sample_bins(const RoiGeom& g, const bf16_t* fb, int H, int W, int C, int P, int n_rot, int K,
                                                int k, bf16_t* out, int part = 0, int nsplit = 1) {
    const int cg = C >> 2;
    const int total = P * P * cg;
    for (int w = threadIdx.x + part * blockDim.x; w < total; w += blockDim.x * nsplit) {
        const int bin = w / cg, c0 = (w - bin * cg) << 2;
        const int py = bin / P, px = bin - py * P;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                const Bil b = bil_setup(y, x, H, W);
                if (!b.valid) continue;
                float v1[4], v2[4], v3[4], v4[4];
                ld4(fb + ((size_t)b.y_low * W + b.x_low) * C + c0, v1);
                ld4(fb + ((size_t)b.y_low * W + b.x_high) * C + c0, v2);
                ld4(fb + ((size_t)b.y_high * W + b.x_low) * C + c0, v3);
                ld4(fb + ((size_t)b.y_high * W + b.x_high) * C + c0, v4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += b.w1 * v1[q] + b.w2 * v2[q] + b.w3 * v3[q] + b.w4 * v4[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] /= g.count;
        for (int r = 0; r < n_rot; ++r)
            st4(out + (((size_t)r * K + k) * P * P + rot_pos(py, px, P, r)) * C + c0, acc);
    }
}

__global__ __launch_bounds__(256) void roi_align_fwd_sep_kernel(RoiLevels L, const float* __restrict__ rois, int K, int C,
                                                                int P, int n_rot, bf16_t* __restrict__ out,
                                                                const int32_t* __restrict__ order) {
    __shared__ float WY[RF_MAXP][RF_MAX], WX[RF_MAXP + 2][RF_MAX];
    __shared__ int ylo[RF_MAXP], yhi[RF_MAXP], xhi[RF_MAXP];
    const int k = roi_of_block(order, K);
    if (k < 0) return;
    const float* roi = rois + 5 * (size_t)k;
    const RoiGeom g = roi_geom(roi, L, P);
    const int H = L.H[g.level], W = L.W[g.level];
    const bf16_t* fb = reinterpret_cast<const bf16_t*>(L.feat[g.level]) + (size_t)g.batch * H * W * C;
    // footprint: every pixel a sample can touch (same bounds as roi_prep_kernel)
    const float end_w = g.start_w + g.bin_w * (float)P, end_h = g.start_h + g.bin_h * (float)P;
    const int x0 = (int)floorf(fminf(fmaxf(fminf(g.start_w, end_w) - 1.f, 0.f), (float)(W - 1)));
    const int x1 = (int)ceilf(fminf(fmaxf(fmaxf(g.start_w, end_w) + 1.f, 0.f), (float)(W - 1)));
    const int y0 = (int)floorf(fminf(fmaxf(fminf(g.start_h, end_h) - 1.f, 0.f), (float)(H - 1)));
    const int y1 = (int)ceilf(fminf(fmaxf(fmaxf(g.start_h, end_h) + 1.f, 0.f), (float)(H - 1)));
    const int Fh = y1 - y0 + 1, Fw = x1 - x0 + 1;
    if (!(g.bin_h >= 1.f && g.bin_w >= 1.f) || Fh > RF_MAX || Fw > RF_MAX || P > RF_MAXP) {   // block-uniform
        roi_sample_bins(g, fb, H, W, C, P, n_rot, K, k, out);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < P * Fh; i += 256) {
        const int p = i / Fh, y = i - p * Fh;
        WY[p][y] = axis_weight(g.start_h, g.bin_h, g.grid_h, p, y0 + y, H);
    }
    for (int i = tid; i < (P + 2) * Fw; i += 256) {
        const int p = i / Fw, x = i - p * Fw;
        WX[p][x] = p < P ? axis_weight(g.start_w, g.bin_w, g.grid_w, p, x0 + x, W) : 0.f;
    }
    __syncthreads();
    if (tid < P) {
        int lo = Fh, hi = -1, xh = -1;
        for (int y = 0; y < Fh; ++y)
            if (WY[tid][y] != 0.f) { lo = min(lo, y); hi = y; }
        for (int x = 0; x < Fw; ++x)
            if (WX[tid][x] != 0.f) xh = x;
        ylo[tid] = lo; yhi[tid] = hi; xhi[tid] = xh;
    }
    __syncthreads();
    const float inv = 1.f / g.count;
    const int cg = C >> 2;
    for (int cb = 0; cb < cg; cb += 64) {
        const bool cact = (cb + lane) < cg;
        const int c0 = (cb + lane) << 2;
        for (int py = wave; py < P; py += 4) {
            const int ya = ylo[py], yb = yhi[py];
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
            int p = 0;
            auto emit = [&]() {
                if (cact) {
                    float v[4] = {a0[0] * inv, a0[1] * inv, a0[2] * inv, a0[3] * inv};
                    for (int r = 0; r < n_rot; ++r)
                        st4(out + (((size_t)r * K + k) * P * P + rot_pos(py, p, P, r)) * C + c0, v);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a0[q] = a1[q]; a1[q] = a2[q]; a2[q] = 0.f; }
                ++p;
            };
            // Column sums of TWO columns at a time, up to four rows each: the (wave-uniform) loads of a pair are all issued
            // before the first one is consumed -- one load per loop trip left the kernel waiting out an L2 round trip per
            // row (the launch was latency-bound at a few percent of the cache bandwidth).
            const int nrow = yb - ya + 1;
            for (int x = 0; x < Fw && p < P; x += 2) {
                float cs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const bool two = x + 1 < Fw;
                if (cact && nrow > 0) {
                    const bf16_t* fp = fb + ((size_t)(y0 + ya) * W + (x0 + x)) * C + c0;
                    for (int yb0 = 0; yb0 < nrow; yb0 += 4) {
                        float t[2][4][4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool in = yb0 + r < nrow;
                            const bf16_t* rp = fp + (size_t)(yb0 + (in ? r : 0)) * W * C;
                            ld4(rp, t[0][r]);
                            ld4(two ? rp + C : rp, t[1][r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float wy = (yb0 + r < nrow) ? WY[py][ya + yb0 + r] : 0.f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) { cs[0][q] += wy * t[0][r][q]; cs[1][q] += wy * t[1][r][q]; }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int xc = x + c;
                    if (xc >= Fw) break;
                    while (p < P && xhi[p] < xc) emit();      // block-uniform: bins whose support ended before this column
                    if (p >= P) break;                        // (bins without any valid sample have xhi = -1)
                    const float w0 = WX[p][xc], w1 = WX[p + 1][xc], w2 = WX[p + 2][xc];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a0[q] += w0 * cs[c][q]; a1[q] += w1 * cs[c][q]; a2[q] += w2 * cs[c][q]; }
                }
            }
            while (p < P) emit();
        }
    }
}

// ---- forward with 16-byte accesses (the shipped 16-bit forward when C % 8 == 0) --------------------------------------
// Round 4's time line of the 8-byte separable kernel (tools/probes/roi_fwd_trace.py: s_memtime stamps per workgroup; ablations
// without loads / without stores / set-up only) said: a workgroup lives 36 .. 100 us, neither its loads nor its stores matter
// (-10 % each when removed), the CUs run 1.8 workgroups deep of the 3-4 the registers allow, and -- with the bench's gt sizes,
// log-uniform 16 .. 160 pixels, i.e. 4 .. 40 pixels on the stride-4 map -- 43 % (P = 7) to 79 % (P = 14) of the positive RoIs
// have bins below one pixel and never reached the separable walk: they ran the sample-order loop, one (bin, 4 channels) item
// per thread and trip with four dependent 8-byte loads, ~50 trips per thread at P = 14.  Three forms now share one launch,
// chosen per RoI (block-uniform):
//   * LDS form (small footprints: up to 24 KiB per channel block of >= 64 channels): the footprint is staged once with
//     global -> LDS copies (all of a pass in flight, no destination registers), then a group of cbch / 8 lanes computes one
//     bin in SAMPLE order from it (wave-uniform sample loops: the grid is per RoI): four 16-byte LDS reads per sample;
//   * separable walk (bins of a pixel or more, larger footprints): a lane owns 8 channels, a half-wave one pixel (lanes 0-31
//     column x, lanes 32-63 column x + 1: one wave access = 1 KiB); the rows of a bin row are loaded in ONE batch whose size is a
//     template parameter selected per bin row (wave-uniform switch), their weights sit in scalar registers, the contraction is
//     v_pk_fma_f32 on channel pairs (~2.2x fewer VALU instructions per window byte than the 8-byte form, whose hipcc code carried
//     row counts in vector registers, 64-bit v_mad addressing and a ds_read per weight in the row loop); both columns of a pair
//     add into a FOUR-bin sliding window with per-half weights (the window base moves by at most one bin between neighbouring
//     columns when bin_w >= 1), halves are summed at emit time with v_permlane32_swap; first / last contributing row and last
//     column of every bin come from ballots instead of serial scans;
//   * the sample-order loop for what is left (large footprints with sub-pixel bins in one direction).
// Lists of up to 4096 RoIs run two workgroups per RoI (bins / bin rows dealt round-robin).  Measured on the three lists of a bench
// step (8192 x 7^2, 2048 x 14^2, 2048 x 7^2 x 4 rotations): 714 -> ~600 us per step; what remains is per-workgroup latency
// (set-up, LDS round trips, uneven lengths), not bytes.
// The fp32 sum order differs from the 8-byte form's and the oracle's (fused multiply-adds, half-wave partial sums): all forms are
// held to the same tolerance against the sample-order kernel and the oracle (bf16 only).
#define RF8_MAXROWS 8
#ifndef RF8_TRACE
#define RF8_TRACE 0       // trace build (tools/probes/roi_fwd_trace.py): `order` is a [K][8] u64 buffer of per-workgroup s_memtime stamps
#endif
#if RF8_TRACE
#define RF8_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(const_cast<int32_t*>(order))[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RF8_NOTE(i, v) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(const_cast<int32_t*>(order))[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)(v); } while (0)
#else
#define RF8_STAMP(i) do { } while (0)
#define RF8_NOTE(i, v) do { } while (0)
#endif
#ifndef RF8_ABL
#define RF8_ABL 0         // timing ablations of tools/probes/roi_fwd_ablate.sh: 1 = prologue only, 2 = no window loads, 3 = no stores
#endif

#ifndef RF8_XCD
#define RF8_XCD 1         // 0: RoIs in launch order (A/B builds)
#endif
template <int NR>
__device__ __forceinline__ void roi_sep8_load(const bf16_t* fp, size_t row_stride, uint4 (&t)[NR ? NR : 1]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (RF8_ABL == 2) t[r] = make_uint4(r, r, r, (unsigned)(size_t)fp);
        else t[r] = *reinterpret_cast<const uint4*>(fp + (size_t)r * row_stride);
    }
}
template <int NR>
__device__ __forceinline__ void roi_sep8_colsum(const uint4 (&t)[NR ? NR : 1], const float (&wy)[RF8_MAXROWS], loft_f32x2 (&cs)[4]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v[8];
        unpack8_16(t[r], v);
        const loft_f32x2 w2 = {wy[r], wy[r]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const loft_f32x2 v2 = {v[2 * q], v[2 * q + 1]};
            cs[q] = __builtin_elementwise_fma(v2, w2, cs[q]);
        }
    }
}

__global__ __launch_bounds__(256) void roi_align_fwd_sep8_kernel(RoiLevels L, const float* __restrict__ rois, int K, int C,
                                                                 int P, int n_rot, bf16_t* __restrict__ out,
                                                                 const int32_t* __restrict__ order, int stage_bytes, int min_cb, int gmax, int nsplit) {
    __shared__ float WY[RF_MAXP][RF_MAX], WX[RF_MAXP + 3][RF_MAX + 2];
    __shared__ int ylo[RF_MAXP], yhi[RF_MAXP], xhi[RF_MAXP];
    extern __shared__ __attribute__((aligned(16))) char stage[];             // footprint of the LDS form: [pixel][cbch channels]
    RF8_STAMP(0);
    // nsplit consecutive workgroups share one RoI (bins / bin rows dealt round-robin): a list of 2048 RoIs is eight workgroups
    // per CU of very uneven length -- finer grains fill the tail
    // XCD-aware order (workgroups are dealt round-robin to the 8 XCDs): workgroup i serves position (i % 8) * (n / 8) + i / 8, so the
    // nsplit parts of one RoI -- the same footprint -- and the RoIs next to it in launch order (sorted by image / level / row strip
    // with `order`) share one L2 instead of being spread over eight
    unsigned bid = blockIdx.x;
#if RF8_XCD
    if ((gridDim.x & 7u) == 0u) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
#endif
    const int part = (int)bid % nsplit, rb = (int)bid / nsplit;
    const int k = RF8_TRACE ? (rb < K ? rb : -1) : roi_of_block(order, K, rb);
    if (k < 0) return;
    const float* roi = rois + 5 * (size_t)k;
    const RoiGeom g = roi_geom(roi, L, P);
    const int H = L.H[g.level], W = L.W[g.level];
    const bf16_t* fb = reinterpret_cast<const bf16_t*>(L.feat[g.level]) + (size_t)g.batch * H * W * C;
    const float end_w = g.start_w + g.bin_w * (float)P, end_h = g.start_h + g.bin_h * (float)P;
    const int x0 = (int)floorf(fminf(fmaxf(fminf(g.start_w, end_w) - 1.f, 0.f), (float)(W - 1)));
    const int x1 = (int)ceilf(fminf(fmaxf(fmaxf(g.start_w, end_w) + 1.f, 0.f), (float)(W - 1)));
    const int y0 = (int)floorf(fminf(fmaxf(fminf(g.start_h, end_h) - 1.f, 0.f), (float)(H - 1)));
    const int y1 = (int)ceilf(fminf(fmaxf(fmaxf(g.start_h, end_h) + 1.f, 0.f), (float)(H - 1)));
    const int Fh = y1 - y0 + 1, Fw = x1 - x0 + 1;
    if (Fh > -100) RF8_STAMP(1);
    const bool tables_ok = Fh <= RF_MAX && Fw <= RF_MAX && P <= RF_MAXP;
    // LDS form: channels per pass = the largest of C, C/2, C/4, .. (>= min_cb, 8 .. 64 lanes per pixel) whose footprint fits
    int cbch = 0;
    if (tables_ok && !(C & (C - 1)) && C >= 64 && C <= 512)
        for (int c = C; c >= min_cb && c >= 64; c >>= 1) {
            const int ppi = 512 / c;                                          // pixels per 1 KiB wave access
            if ((Fh * Fw + ppi - 1) / ppi * 1024 <= stage_bytes) { cbch = c; break; }
        }
    const bool stream_ok = tables_ok && g.bin_h >= 1.f && g.bin_w >= 1.f;
    // (bins of a pixel or more with many samples each: the separable walk reads every window pixel once per bin row, the LDS
    //  form below once per sample corner)
    if (stream_ok && g.grid_h * g.grid_w > gmax) cbch = 0;
    RF8_NOTE(5, (cbch ? cbch : stream_ok ? 1 : 0) | (Fh << 16) | (Fw << 24));
#if RF8_TRACE
    if (threadIdx.x == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); RF8_NOTE(6, hw | ((unsigned long long)xcc << 32)); }
#endif
    if (!cbch && !stream_ok) {                                                // block-uniform
        roi_sample_bins(g, fb, H, W, C, P, n_rot, K, k, out, part, nsplit);
#if RF8_TRACE
        __syncthreads();
        RF8_STAMP(4);
#endif
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float inv = 1.f / g.count;
    if (cbch) {
        // ---- LDS form: the footprint is staged ONCE (every load of a pass in flight together and no destination registers:
        // global -> LDS copies of 1 KiB per wave access), then a group of cbch / 8 lanes computes one bin from it in SAMPLE order
        // (the grid is the same for every bin of a RoI, so the sample loops are wave-uniform): four 16-byte LDS reads per sample.
        // This is the path of the small RoIs -- bins below one pixel, one sample per bin: building roofs of 16 .. 56 pixels on
        // the stride-4 map, 40 .. 80 % of the positives -- which the sample-order loop served with four dependent 8-byte global
        // loads per bin and thread, ~50 round trips per workgroup at P = 14.
        RF8_STAMP(2);
        const int lsh = 31 - __builtin_clz(cbch >> 3);          // log2(lanes per bin)
        const int lpp = 1 << lsh, ppi = 64 >> lsh;
        const int pl = lane >> lsh, cl = lane & (lpp - 1);
        const int npx = Fh * Fw, nacc = (npx + ppi - 1) / ppi, nb = P * P;
        const unsigned pdiv = (65536u + (unsigned)P - 1u) / (unsigned)P;          // b / P for b < 256, P <= 16 as a multiply-high
        for (int c0 = 0; c0 < C; c0 += cbch) {
            if (c0) __syncthreads();                            // the previous pass's reads are done
            for (int i = wave; i < nacc; i += 4) {
                int pix = i * ppi + pl;
                pix = pix < npx ? pix : npx - 1;                // (the last access may repeat the last pixel into the slack)
                const int y = pix / Fw, x = pix - y * Fw;
                const bf16_t* src = fb + ((size_t)(y0 + y) * W + x0 + x) * C + c0 + cl * 8;
                if (RF8_ABL != 2)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage + (size_t)i * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c0 == 0) RF8_STAMP(3);
            const char* lbase = stage + cl * 16;
            for (int b0 = (wave + 4 * part) * ppi; b0 < nb; b0 += 4 * nsplit * ppi) {
                const int b = b0 + pl;
                const bool act = b < nb;
                const int py = act ? (int)(((unsigned)b * pdiv) >> 16) : 0, px = act ? b - py * P : 0;
                loft_f32x2 v2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    const float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                        const Bil bl = bil_setup(y, x, H, W);
                        // footprint coordinates (every corner of a valid sample lies inside the footprint; invalid ones read pixel 0 at weight 0)
                        const int ya = bl.valid ? bl.y_low - y0 : 0, yb = bl.valid ? bl.y_high - y0 : 0;
                        const int xa = bl.valid ? bl.x_low - x0 : 0, xb = bl.valid ? bl.x_high - x0 : 0;
                        const float w1 = bl.valid ? bl.w1 : 0.f, w2 = bl.valid ? bl.w2 : 0.f, w3 = bl.valid ? bl.w3 : 0.f, w4 = bl.valid ? bl.w4 : 0.f;
                        const uint4 t1 = *reinterpret_cast<const uint4*>(lbase + ((size_t)(ya * Fw + xa) << (lsh + 4)));
                        const uint4 t2 = *reinterpret_cast<const uint4*>(lbase + ((size_t)(ya * Fw + xb) << (lsh + 4)));
                        const uint4 t3 = *reinterpret_cast<const uint4*>(lbase + ((size_t)(yb * Fw + xa) << (lsh + 4)));
                        const uint4 t4 = *reinterpret_cast<const uint4*>(lbase + ((size_t)(yb * Fw + xb) << (lsh + 4)));
                        float f1[8], f2[8], f3[8], f4[8];
                        unpack8_16(t1, f1); unpack8_16(t2, f2); unpack8_16(t3, f3); unpack8_16(t4, f4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            loft_f32x2 a = v2[q];
                            a = __builtin_elementwise_fma(loft_f32x2{f1[2 * q], f1[2 * q + 1]}, loft_f32x2{w1, w1}, a);
                            a = __builtin_elementwise_fma(loft_f32x2{f2[2 * q], f2[2 * q + 1]}, loft_f32x2{w2, w2}, a);
                            a = __builtin_elementwise_fma(loft_f32x2{f3[2 * q], f3[2 * q + 1]}, loft_f32x2{w3, w3}, a);
                            v2[q] = __builtin_elementwise_fma(loft_f32x2{f4[2 * q], f4[2 * q + 1]}, loft_f32x2{w4, w4}, a);
                        }
                    }
                }
                if (act && !(RF8_ABL == 3 && v2[0][0] != 1234.5f)) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = v2[q][0] * inv; v[2 * q + 1] = v2[q][1] * inv; }
                    const uint4 pk = pack8_16(v);
                    for (int r = 0; r < n_rot; ++r)
                        *reinterpret_cast<uint4*>(out + (((size_t)r * K + k) * P * P + rot_pos(py, px, P, r)) * C + c0 + cl * 8) = pk;
                }
            }
        }
#if RF8_TRACE
        __syncthreads();
        RF8_STAMP(4);
#endif
        return;
    }
    for (int i = tid; i < P * Fh; i += 256) {
        const int p = i / Fh, y = i - p * Fh;
        WY[p][y] = axis_weight(g.start_h, g.bin_h, g.grid_h, p, y0 + y, H);
    }
    const int Fw2 = Fw + 2;                                   // two zero columns: half 1 of an odd last pair reads column Fw
    for (int i = tid; i < (P + 3) * Fw2; i += 256) {
        const int p = i / Fw2, x = i - p * Fw2;
        WX[p][x] = (p < P && x < Fw) ? axis_weight(g.start_w, g.bin_w, g.grid_w, p, x0 + x, W) : 0.f;
    }
    __syncthreads();
    for (int p = wave; p < P; p += 4) {
        const unsigned long long my = __ballot(lane < Fh && WY[p][lane < Fh ? lane : 0] != 0.f);
        const unsigned long long mx = __ballot(lane < Fw && WX[p][lane < Fw ? lane : 0] != 0.f);
        if (lane == 0) {
            ylo[p] = my ? __builtin_ctzll(my) : Fh;
            yhi[p] = my ? 63 - __builtin_clzll(my) : -1;
            xhi[p] = mx ? 63 - __builtin_clzll(mx) : -1;
        }
    }
    __syncthreads();
    if (RF8_ABL == 1) { if (ylo[0] == 12345) out[0] = 0; return; }
    RF8_STAMP(2);
    const int h = lane >> 5, cl = lane & 31;
    const int cg8 = C >> 3;
    const size_t row_stride = (size_t)W * C;
    const int xhi_v = lane < P ? xhi[lane] : 0x7fffffff;      // lane p holds xhi[p]: read back with v_readlane at the scalar p
    for (int cb = 0; cb < cg8; cb += 32) {
        const bool cact = cb + cl < cg8;
        const int c0 = (cact ? cb + cl : 0) << 3;
        for (int py = wave + 4 * part; py < P; py += 4 * nsplit) {
            const int ya = __builtin_amdgcn_readfirstlane(ylo[py]);
            const int nrow = __builtin_amdgcn_readfirstlane(yhi[py]) - ya + 1;
            loft_f32x2 a[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[j][q] = loft_f32x2{0.f, 0.f};
            int p = 0;
            auto emit = [&]() {
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned u = __float_as_uint(a[0][q][e]);
                        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lower half, upper half} in every lane
                        v[2 * q + e] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * inv;
                    }
                if (cact && !(RF8_ABL == 3 && v[0] != 1234.5f)) {
                    const uint4 pk = pack8_16(v);
                    for (int r = h; r < n_rot; r += 2)       // half 0 stores rotations 0, 2; half 1 rotations 1, 3
                        *reinterpret_cast<uint4*>(out + (((size_t)r * K + k) * P * P + rot_pos(py, p, P, r)) * C + c0) = pk;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[0][q] = a[1][q]; a[1][q] = a[2][q]; a[2][q] = a[3][q]; a[3][q] = loft_f32x2{0.f, 0.f}; }
                ++p;
            };
            auto walk = [&](auto nr_tag) {
                constexpr int NR = decltype(nr_tag)::value;          // rows per batch; NR == 0: no contributing row at all
                float wy[RF8_MAXROWS];
#pragma unroll
                for (int r = 0; r < RF8_MAXROWS; ++r)
                    wy[r] = (r < NR && r < nrow) ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(WY[py][ya + r]))) : 0.f;
                const bf16_t* colp = fb + ((size_t)(y0 + (NR ? ya : 0)) * W + x0) * C + c0;      // column x of the first contributing row
                const int hC = h ? C : 0;
                // the loads of pair x + 2 are requested before pair x is consumed (batches of <= 4 rows: a second set of
                // destination registers costs less than the round trip it hides; taller batches keep one set)
                constexpr bool PRE = NR > 0 && NR <= 4;
                uint4 tc[NR ? NR : 1], tn[NR ? NR : 1];
                if constexpr (PRE) roi_sep8_load<NR>(colp + (1 < Fw ? hC : 0), row_stride, tc);
                for (int x = 0; x < Fw; x += 2, colp += 2 * C) {
                    while (p < P && __builtin_amdgcn_readlane(xhi_v, p) < x) emit();      // bins whose support ended before column x
                    if (p >= P) break;
                    loft_f32x2 cs[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
                    if constexpr (NR > 0) {
                        const bf16_t* fp = colp + (x + 1 < Fw ? hC : 0);    // an odd last column: half 1 re-reads it (its weights are zero)
                        if constexpr (PRE) {
                            if (x + 2 < Fw) roi_sep8_load<NR>(colp + 2 * C + (x + 3 < Fw ? hC : 0), row_stride, tn);
                        } else {
                            roi_sep8_load<NR>(fp, row_stride, tc);
                        }
                        roi_sep8_colsum<NR>(tc, wy, cs);
                        if constexpr (PRE) {
#pragma unroll
                            for (int r = 0; r < NR; ++r) tc[r] = tn[r];
                        }
                        if (NR == RF8_MAXROWS)
                            for (int r0 = RF8_MAXROWS; r0 < nrow; r0 += RF8_MAXROWS) {   // bins taller than 8 + 2 rows: further batches,
                                float wz[RF8_MAXROWS];                                    // rows past the end re-read the last one at weight 0
#pragma unroll
                                for (int r = 0; r < RF8_MAXROWS; ++r) wz[r] = r0 + r < nrow ? WY[py][ya + r0 + r] : 0.f;
                                const int last = nrow - 1 - r0;
                                uint4 t[RF8_MAXROWS];
#pragma unroll
                                for (int r = 0; r < RF8_MAXROWS; ++r)
                                    t[r] = *reinterpret_cast<const uint4*>(fp + (size_t)(r0 + (r < last ? r : last)) * row_stride);
#pragma unroll
                                for (int r = 0; r < RF8_MAXROWS; ++r) {
                                    float v[8];
                                    unpack8_16(t[r], v);
                                    const loft_f32x2 w2 = {wz[r], wz[r]};
#pragma unroll
                                    for (int q = 0; q < 4; ++q) cs[q] = __builtin_elementwise_fma(loft_f32x2{v[2 * q], v[2 * q + 1]}, w2, cs[q]);
                                }
                            }
                    }
                    const int xw = x + h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float w = WX[p + j][xw];
                        const loft_f32x2 w2 = {w, w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) a[j][q] = __builtin_elementwise_fma(cs[q], w2, a[j][q]);
                    }
                }
                while (p < P) emit();
            };
            switch (nrow <= 0 ? 0 : (nrow < RF8_MAXROWS ? nrow : RF8_MAXROWS)) {
                case 0: walk(std::integral_constant<int, 0>{}); break;
                case 1: walk(std::integral_constant<int, 1>{}); break;
                case 2: walk(std::integral_constant<int, 2>{}); break;
                case 3: walk(std::integral_constant<int, 3>{}); break;
                case 4: walk(std::integral_constant<int, 4>{}); break;
                case 5: walk(std::integral_constant<int, 5>{}); break;
                case 6: walk(std::integral_constant<int, 6>{}); break;
                case 7: walk(std::integral_constant<int, 7>{}); break;
                default: walk(std::integral_constant<int, 8>{}); break;
            }
        }
    }
#if RF8_TRACE
    __syncthreads();
    RF8_STAMP(4);
#endif
}

// ---- backward: tile-owner gather -- no atomics at all, neither in HBM nor in LDS ----------------------
// One workgroup OWNS an 8x8-pixel tile of one image's gradient map at one pyramid level.  Wave w owns tile rows
// 2w, 2w+1 and lane l owns channels 4l..4l+3, so the whole tile lives in 64 accumulator VGPRs per lane.  For every
// RoI of this (image, level) whose footprint touches the tile (16-byte records from a prepass, RoIs of one image
// found by binary search) the bilinear scatter is turned into a SEPARABLE gather:
//     grad[y,x,:] += sum_{py,px} gout[py,px,:] / count * WY[py][y] * WX[px][x]
// with WY[py][y] = sum over the bin's sample rows of the (clamped) hat weight of row y -- exactly the weights
// mmcv's roi_align backward applies sample by sample, regrouped.  The two 1-D tables (<= 14x8 each) are rebuilt in
// LDS per (RoI, tile); each needed gout row is read once per wave as one coalesced 512-byte access.
// The tile is written (or added to) in HBM exactly once.
#define RB_TILE 8
#define RB_LIST 256
#define RB_MAXP 14

// rec[K]: (image, level | empty << 8, x0 | x1 << 16, y0 | y1 << 16) footprints; behind them geo[2K]: the RoI's sampling geometry
// (start_h, start_w, bin_h, bin_w | grid_h, grid_w as int bits, count, 1 / count) so that the per-(RoI, tile) tables read 32 bytes instead
// of redoing roi_geom's square root, logarithm and divisions for every tile the RoI touches.  48 bytes of workspace per RoI.
__global__ void roi_prep_kernel(RoiLevels L, const float* __restrict__ rois, int K, int P, int4* __restrict__ rec) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* roi = rois + 5 * (size_t)k;
    const RoiGeom g = roi_geom(roi, L, P);
    float4* geo = reinterpret_cast<float4*>(rec + K);
    geo[2 * k] = make_float4(g.start_h, g.start_w, g.bin_h, g.bin_w);
    geo[2 * k + 1] = make_float4(__int_as_float(g.grid_h), __int_as_float(g.grid_w), g.count, 1.f / g.count);
    const int H = L.H[g.level], W = L.W[g.level];
    const float end_w = g.start_w + g.bin_w * (float)P, end_h = g.start_h + g.bin_h * (float)P;
    const float lo_x = fminf(g.start_w, end_w) - 1.f, hi_x = fmaxf(g.start_w, end_w) + 1.f;
    const float lo_y = fminf(g.start_h, end_h) - 1.f, hi_y = fmaxf(g.start_h, end_h) + 1.f;
    const int x0 = (int)floorf(fminf(fmaxf(lo_x, 0.f), (float)(W - 1))), x1 = (int)ceilf(fminf(fmaxf(hi_x, 0.f), (float)(W - 1)));
    const int y0 = (int)floorf(fminf(fmaxf(lo_y, 0.f), (float)(H - 1))), y1 = (int)ceilf(fminf(fmaxf(hi_y, 0.f), (float)(H - 1)));
    const int empty = (g.grid_h <= 0 || g.grid_w <= 0) ? 1 : 0;   // zero-size RoI: no samples, no gradient
    rec[k] = make_int4(g.batch, g.level | (empty << 8), x0 | (x1 << 16), y0 | (y1 << 16));
}

// 1-D weight of tile row/col `pix` from the samples of bin `p`: same clamping rules as bil_setup
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int pix, int size) {
    float w = 0.f;
    for (int i = 0; i < grid; ++i) {
        float v = start + p * bin + ((float)i + .5f) * bin / (float)grid;
        if (v < -1.0f || v > (float)size) continue;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else hi = lo + 1;
        const float l = v - (float)lo, h = 1.f - l;
        if (lo == pix) w += h;
        if (hi == pix) w += l;
    }
    return w;
}

// GT: type of the gradient map (fp32, or bf16 written directly: every pixel is written exactly once from fp32 registers, so
// no fp32 staging map + cast pass is needed when the features are bf16)
template <typename T, typename GT>
__global__ __launch_bounds__(256) void roi_align_bwd_tile_kernel(RoiLevels L, int level, const float* __restrict__ rois,
                                                                 int K, int C, int P, int n_rot,
                                                                 const T* __restrict__ gout, GT* __restrict__ grad,
                                                                 int accumulate, const int4* __restrict__ rec, int sorted) {
    __shared__ int list[RB_LIST];
    __shared__ int wcnt[4];
    __shared__ int range[2];
    __shared__ float WY[RB_MAXP][RB_TILE], WX[RB_MAXP][RB_TILE];
    __shared__ int rowany[RB_MAXP][4], colany[RB_MAXP];
    const int H = L.H[level], W = L.W[level];
    const int tx0 = blockIdx.x * RB_TILE, ty0 = blockIdx.y * RB_TILE, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = C >> 2;
    if (tid < 2) {
        int lo = 0, hi = K;
        const int key = b + tid;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[mid].x < key) lo = mid + 1; else hi = mid; }
        range[tid] = lo;
    }
    __syncthreads();
    const int kbeg = sorted ? range[0] : 0, kend = sorted ? range[1] : K;
    for (int cb = 0; cb < cg; cb += 64) {      // 256 channels per pass (one pass for the FPN's C = 256)
        const int c0 = (cb + lane) << 2;
        const bool cact = (cb + lane) < cg;
        float acc[16][4];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
        for (int base = kbeg; base < kend; base += RB_LIST) {
            // ---- deterministic compaction of the RoIs that touch this tile
            const int k = base + tid;
            bool hit = false;
            if (k < kend) {
                const int4 r = rec[k];
                if (r.x == b && r.y == level) {   // (the empty flag lives in bits 8+ of r.y -> empty RoIs never match)
                    const int x0 = r.z & 0xffff, x1 = r.z >> 16, y0 = r.w & 0xffff, y1 = r.w >> 16;
                    hit = x1 >= tx0 && x0 < tx0 + RB_TILE && y1 >= ty0 && y0 < ty0 + RB_TILE;
                }
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            int off = 0;
            for (int w2 = 0; w2 < wave; ++w2) off += wcnt[w2];
            const int n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            if (hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = k;
            __syncthreads();
            for (int li = 0; li < n; ++li) {
                const int kk = list[li];
                const RoiGeom g = roi_geom(rois + 5 * (size_t)kk, L, P);
                if (tid < P * RB_TILE) {
                    const int p = tid / RB_TILE, pix = tid % RB_TILE;
                    WY[p][pix] = axis_weight(g.start_h, g.bin_h, g.grid_h, p, ty0 + pix, H);
                } else if (tid >= 128 && tid < 128 + P * RB_TILE) {
                    const int t = tid - 128, p = t / RB_TILE, pix = t % RB_TILE;
                    WX[p][pix] = axis_weight(g.start_w, g.bin_w, g.grid_w, p, tx0 + pix, W);
                }
                __syncthreads();
                if (tid < P) {
                    int any = 0;
                    for (int x = 0; x < RB_TILE; ++x) any |= (WX[tid][x] != 0.f);
                    colany[tid] = any;
                } else if (tid >= 64 && tid < 64 + P * 4) {
                    const int t = tid - 64, p = t >> 2, w2 = t & 3;
                    rowany[p][w2] = (WY[p][2 * w2] != 0.f) | (WY[p][2 * w2 + 1] != 0.f);
                }
                __syncthreads();
                const float inv = 1.f / g.count;
                for (int py = 0; py < P; ++py) {
                    if (!rowany[py][wave]) continue;
                    const float wy0 = WY[py][2 * wave] * inv, wy1 = WY[py][2 * wave + 1] * inv;
                    for (int px = 0; px < P; ++px) {
                        if (!colany[px]) continue;
                        float gv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (cact)
                            for (int r = 0; r < n_rot; ++r) {
                                float t[4];
                                ld4(gout + (((size_t)r * K + kk) * P * P + rot_pos(py, px, P, r)) * C + c0, t);
#pragma unroll
                                for (int q = 0; q < 4; ++q) gv[q] += t[q];
                            }
#pragma unroll
                        for (int x = 0; x < RB_TILE; ++x) {
                            const float wx = WX[px][x];
                            const float a0 = wy0 * wx, a1 = wy1 * wx;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc[x][q] += a0 * gv[q];
                                acc[8 + x][q] += a1 * gv[q];
                            }
                        }
                    }
                }
                __syncthreads();   // tables are rewritten by the next RoI
            }
        }
        // ---- flush: wave w writes its two rows, every pixel one coalesced 1 KiB (fp32) access
        if (cact) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int y = ty0 + 2 * wave + (i >> 3), x = tx0 + (i & 7);
                if (y >= H || x >= W) continue;
                GT* gp = grad + (((size_t)b * H + y) * W + x) * C + c0;
                float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                if (accumulate) {
                    float o[4];
                    ld4(gp, o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += o[q];
                }
                st4(gp, v);
            }
        }
    }
}

// ---- backward on the matrix cores (bf16 training path, C == 256) ---------------------------------------------------
// Same tile-owner scheme, but the per-(RoI, tile) scatter is one small GEMM instead of 64 VALU FMAs per bin and lane:
//     grad_tile[c][pix] += sum_k  gout[bin_k][c] * A[bin_k][pix],   A[bin][pix] = WY[py][y] * WX[px][x] / count
// over the (<= 64 per chunk) bins that overlap the 8x8 tile.  gout rows are staged bin-major in LDS (summed over the FOA
// rotations), A is built bin-major next to them as a bf16 hi + lo pair (16 mantissa bits: the fp32 weights of the scalar
// form to 1e-5), both operands are read with the LDS transpose reader of the weight-gradient kernel
// (ds_read_tr16_b64: "8 consecutive k for my column"), v_mfma_f32_32x32x16_bf16 accumulates in fp32.
// Wave w owns channels 64w..64w+63 of all 64 pixels (2 x 2 tiles of 32 x 32 -> 64 accumulator registers).  Per pair
// this is ~2 x 2 x 2 x ceil(K/16) MFMAs instead of K x 64 FMAs per lane; the VALU form spent 1 ms on the P2 level
// of the 8192-RoI bbox extractor alone.
typedef __attribute__((ext_vector_type(8))) short rbf16x8;
typedef __attribute__((ext_vector_type(16))) float rf32x16;
typedef __attribute__((ext_vector_type(4))) short rs16x4;

__device__ __forceinline__ int rwswz(int row, int q) { return q ^ ((row & 3) << 2); }

template <int RB>
__device__ __forceinline__ rbf16x8 roi_tr_frag(const char* tile, int kbase, int col0, int lane) {
    // lane l -> column col0 + (l&31), rows (k) kbase + 8*(l>>5) + 0..7 of a row-major [k][RB/2] bf16 tile
    const int il = lane & 15, gl = lane >> 4;
    const int col = col0 + 16 * (gl & 1) + (il & 3) * 4;
    const int r0 = kbase + 8 * (gl >> 1) + (il >> 2);
    const int r1 = r0 + 4;
    const char* p0 = tile + r0 * RB + rwswz(r0, col >> 3) * 16 + (col & 7) * 2;
    const char* p1 = tile + r1 * RB + rwswz(r1, col >> 3) * 16 + (col & 7) * 2;
    rs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)p0);
    rs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)p1);
    rbf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int RB>
__device__ __forceinline__ const char* roi_tr_frag_ptr(const char* tile, int col0, int lane) {     // roi_tr_frag's p0 at kbase = 0
    const int il = lane & 15, gl = lane >> 4;
    const int col = col0 + 16 * (gl & 1) + (il & 3) * 4;
    const int r0 = 8 * (gl >> 1) + (il >> 2);
    return tile + r0 * RB + rwswz(r0, col >> 3) * 16 + (col & 7) * 2;
}

template <int RB, int KS>
__device__ __forceinline__ rbf16x8 roi_tr_frag_at(const char* p0) {                                // kbase = KS (a multiple of 16)
    static_assert(KS % 16 == 0, "the row swizzle repeats every 4 rows");
    rs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)(p0 + KS * RB));
    rs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)(p0 + (KS + 4) * RB));
    rbf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ int nth_set_bit(unsigned m, int n) {      // position of the n-th (0-based) set bit of m
    for (int i = 0; i < n; ++i) m &= m - 1u;
    return __builtin_ctz(m);
}

#ifndef RBM_ABL
#define RBM_ABL 0         // timing ablations of the backward kernel (results WRONG): 1 no MFMAs (the compiler then drops the staging
#endif                    // too), 2 no operand build, 3 no gout staging, 5 no pairs at all (tools/probes/roi_bwd_ablate.sh)
#ifndef RBM_WAVES
#define RBM_WAVES 4
#endif
#ifndef RBM_XCD
#define RBM_XCD 1         // 0: tiles in launch order (A/B builds)
#endif
#define RBM_CHUNK 32      // bins per operand chunk: 24 KiB of LDS per block -> 6 blocks per CU hide the staging latency

// Up to RBM_SETS RoI lists (the three extractors of the LOFT head share one pyramid) scatter into the same maps in ONE
// launch: the tile owner walks list after list into the same fp32 accumulators and writes each pixel once.
#define RBM_SETS 3
struct RoiBwdSets {
    const float* rois[RBM_SETS];
    const bf16_t* gout[RBM_SETS];
    const int4* rec[RBM_SETS];
    int K[RBM_SETS], P[RBM_SETS], n_rot[RBM_SETS], sorted[RBM_SETS];
    int n;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RBM_WAVES))) void roi_align_bwd_mfma_kernel(RoiLevels L, int level, RoiBwdSets S,
                                                                 bf16_t* __restrict__ grad, int accumulate) {
    constexpr int C = 256;
    __shared__ __attribute__((aligned(16))) char gbuf[RBM_CHUNK * 512];      // [bin][256 ch] bf16, 16-byte chunks swizzled
    __shared__ __attribute__((aligned(16))) char abuf[RBM_CHUNK * 256];      // [bin][64 px hi | 64 px lo] bf16
    __shared__ int list[RB_LIST];
    __shared__ int wcnt[4];
    __shared__ int range[2];
    __shared__ float WY[2][RB_MAXP][RB_TILE], WX[2][RB_MAXP][RB_TILE];       // double-buffered per pair
    __shared__ unsigned rnpt[16];
    __shared__ unsigned anyb[2][4];                                          // per wave: which of its 8 bins have any weight
    const int H = L.H[level], W = L.W[level];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so consecutive workgroup ids -- horizontally adjacent
    // tiles, which share the gout rows of every RoI that straddles them -- land on 8 different L2s.  Workgroup L serves tile
    // (L % 8) * (total / 8) + L / 8 instead: each XCD walks one contiguous eighth of the (image, tile row, tile) order, and the tiles
    // it has in flight at one time (~128) are a few adjacent tile rows of one image.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#if RBM_XCD
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z, lin = bx + gx * (by + gy * bz);
        if ((total & 7u) == 0u) {
            const unsigned t = (lin & 7u) * (total >> 3) + (lin >> 3);
            bx = (int)(t % gx); by = (int)((t / gx) % gy); bz = (int)(t / (gx * gy));
        }
    }
#endif
    const int tx0 = bx * RB_TILE, ty0 = by * RB_TILE, b = bz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < RBM_CHUNK * 512 / 16; i += 256) reinterpret_cast<uint4*>(gbuf)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < RBM_CHUNK * 256 / 16; i += 256) reinterpret_cast<uint4*>(abuf)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 16) rnpt[tid] = tid > 1 ? (65536u + (unsigned)tid - 1u) / (unsigned)tid : 65536u;   // ceil(2^16 / n): x / n for x < 256
    // fragment addresses of the transposing reads: k-steps start at multiples of 16 rows, so the row swizzle does not depend on
    // the step and a lane's six addresses are computed once (roi_tr_frag_at adds the step as an immediate)
    const char* gfp[2];
    const char* afp[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) gfp[i] = roi_tr_frag_ptr<512>(gbuf, wave * 64 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        afp[j] = roi_tr_frag_ptr<256>(abuf, j * 32, lane);
        afp[2 + j] = roi_tr_frag_ptr<256>(abuf, 64 + j * 32, lane);
    }
    rf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* rois = nullptr;
    const bf16_t* gout = nullptr;
    const int4* rec = nullptr;
    int K = 0, P = 0, n_rot = 1;

    // 1-D weight tables of one RoI for this tile + which bin rows / columns have any weight in it (wave ballots: waves 0-1
    // hold the 8 tile rows of bins 0-7 / 8-13, waves 2-3 the tile columns)
    auto tables = [&](int kk, int buf) {
        const float4* geo = reinterpret_cast<const float4*>(rec + K);
        const float4 g0 = geo[2 * kk], g1 = geo[2 * kk + 1];
        RoiGeom g;
        g.start_h = g0.x; g.start_w = g0.y; g.bin_h = g0.z; g.bin_w = g0.w;
        g.grid_h = __float_as_int(g1.x); g.grid_w = __float_as_int(g1.y); g.count = g1.z;
        const int t = tid & 127;
        const int p = t >> 3, pix = t & 7;
        float w = 0.f;
        if (p < P) {
            if (tid < 128) {
                w = axis_weight(g.start_h, g.bin_h, g.grid_h, p, ty0 + pix, H);
                WY[buf][p][pix] = w;
            } else {
                w = axis_weight(g.start_w, g.bin_w, g.grid_w, p, tx0 + pix, W);
                WX[buf][p][pix] = w;
            }
        }
        // bit p of the wave's mask = "bin 8*(wave&1) + p has weight on some row / column of this tile": byte p of the ballot is
        // non-zero (wave-uniform -> scalar ALU; the nine-shift fold + multiply gathers bit 0 of every byte into the top byte)
        unsigned long long bt = __ballot(w != 0.f);
        bt |= bt >> 4; bt |= bt >> 2; bt |= bt >> 1;
        bt &= 0x0101010101010101ull;
        if (lane == 0) anyb[buf][wave] = (unsigned)((bt * 0x0102040810204080ull) >> 56);
        return g1.w;
    };

#pragma unroll 1
    for (int si = 0; si < S.n; ++si) {
    rois = S.rois[si]; gout = S.gout[si]; rec = S.rec[si];
    K = S.K[si]; P = S.P[si]; n_rot = S.n_rot[si];
    __syncthreads();                         // (previous list: its last reads of range[] and the tables are done)
    if (tid < 2) {
        int lo = 0, hi = K;
        const int key = b + tid;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[mid].x < key) lo = mid + 1; else hi = mid; }
        range[tid] = lo;
    }
    __syncthreads();
    const int kbeg = S.sorted[si] ? range[0] : 0, kend = S.sorted[si] ? range[1] : K;
    for (int base = kbeg; base < kend; base += RB_LIST) {
        // ---- deterministic compaction of the RoIs that touch this tile
        const int k = base + tid;
        bool hit = false;
        if (k < kend) {
            const int4 r = rec[k];
            if (r.x == b && r.y == level) {   // (the empty flag lives in bits 8+ of r.y -> empty RoIs never match)
                const int x0 = r.z & 0xffff, x1 = r.z >> 16, y0 = r.w & 0xffff, y1 = r.w >> 16;
                hit = x1 >= tx0 && x0 < tx0 + RB_TILE && y1 >= ty0 && y0 < ty0 + RB_TILE;
            }
        }
        const unsigned long long bal = __ballot(hit);
        __syncthreads();                     // (previous batch: its last fragment reads and list reads are done)
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w2 = 0; w2 < wave; ++w2) off += wcnt[w2];
        const int n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        float inv = 0.f;
        if (n > 0) inv = tables(list[0], 0);
#if RBM_ABL == 5
        if (n > 0) acc[0][0][0] += inv;          // (timing ablation: the scan alone)
        for (int li = 0; li < 0; ++li) {
#else
        for (int li = 0; li < n; ++li) {
#endif
            const int kk = list[li];
            const int buf = li & 1;
            __syncthreads();                 // tables of this pair are ready; the previous pair's fragment reads are done
            const unsigned rmask = anyb[buf][0] | (anyb[buf][1] << 8), cmask = anyb[buf][2] | (anyb[buf][3] << 8);
            const int npy = __popc(rmask), npx = __popc(cmask);
            // ab / npx for ab < 256, npx <= 14 as a multiply-high: exact (the error term stays below 1/256 of a unit)
            const unsigned rnp = rnpt[npx];
            const int ka = npy * npx;            // active bins of this (RoI, tile) pair
            const float inv_cur = inv;

            for (int kc = 0; kc < ka; kc += RBM_CHUNK) {          // one chunk unless P = 14 with small bins
                const int kn = min(RBM_CHUNK, ka - kc);
                const int kpad = (kn + 15) & ~15;
                if (kc > 0) __syncthreads();                      // the previous chunk's fragment reads are done
                // ---- A[bin][pix]: one (bin, tile row) = 8 pixels = one 16-byte chunk of hi and one of lo per task
#pragma unroll 1
                for (int t = tid; t < (RBM_ABL == 2 ? 0 : kpad * 8); t += 256) {
                    const int kb = t >> 3, y = t & 7;
                    uint32_t hi4[4] = {0, 0, 0, 0}, lo4[4] = {0, 0, 0, 0};
                    if (kb < kn) {
                        const int ab = kc + kb;
                        const int qy = (int)(((unsigned)ab * rnp) >> 16);
                        const int py = nth_set_bit(rmask, qy), px = nth_set_bit(cmask, ab - qy * npx);
                        const float wy = WY[buf][py][y] * inv_cur;
#pragma unroll
                        for (int x = 0; x < 8; ++x) {
                            const float a = wy * WX[buf][px][x];
                            const bf16_t h = f32_to_bf16(a);
                            const bf16_t l = f32_to_bf16(a - bf16_to_f32(h));
                            hi4[x >> 1] |= (uint32_t)h << (16 * (x & 1));
                            lo4[x >> 1] |= (uint32_t)l << (16 * (x & 1));
                        }
                    }
                    *reinterpret_cast<uint4*>(abuf + kb * 256 + rwswz(kb, y) * 16) = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
                    *reinterpret_cast<uint4*>(abuf + kb * 256 + rwswz(kb, 8 + y) * 16) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
                }
                // ---- gout rows of the active bins (FOA: the four rotations summed in fp32, rounded once)
#pragma unroll 1
                for (int t = tid; t < (RBM_ABL == 3 ? 0 : kn * 32); t += 256) {
                    const int kb = t >> 5, q = t & 31;
                    const int ab = kc + kb;
                    const int qy = (int)(((unsigned)ab * rnp) >> 16);
                    const int py = nth_set_bit(rmask, qy), px = nth_set_bit(cmask, ab - qy * npx);
                    uint4 v;
                    if (n_rot == 1) {
                        v = *reinterpret_cast<const uint4*>(gout + ((size_t)kk * P * P + py * P + px) * C + q * 8);
                    } else {
                        // all four rotations' rows are requested before the first is summed (a dependent load per rotation made
                        // an FOA pair five times as expensive as a bbox pair); same r = 0..3 fp32 summation order
                        float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        uint4 rv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            rv[r] = *reinterpret_cast<const uint4*>(gout + (((size_t)r * K + kk) * P * P + rot_pos(py, px, P, r)) * C + q * 8);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float tv[8];
                            unpack8_16(rv[r], tv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) sacc[e] += tv[e];
                        }
                        v.x = (uint32_t)f32_to_bf16(sacc[0]) | ((uint32_t)f32_to_bf16(sacc[1]) << 16);
                        v.y = (uint32_t)f32_to_bf16(sacc[2]) | ((uint32_t)f32_to_bf16(sacc[3]) << 16);
                        v.z = (uint32_t)f32_to_bf16(sacc[4]) | ((uint32_t)f32_to_bf16(sacc[5]) << 16);
                        v.w = (uint32_t)f32_to_bf16(sacc[6]) | ((uint32_t)f32_to_bf16(sacc[7]) << 16);
                    }
                    *reinterpret_cast<uint4*>(gbuf + kb * 512 + rwswz(kb, q) * 16) = v;
                }
                __syncthreads();
#define RBM_KSTEP(KS)                                                                                         \
                do {                                                                                          \
                    rbf16x8 gf[2], xh[2], xl[2];                                                              \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) gf[i] = roi_tr_frag_at<512, KS>(gfp[i]);    \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
                        xh[j] = roi_tr_frag_at<256, KS>(afp[j]);                                              \
                        xl[j] = roi_tr_frag_at<256, KS>(afp[2 + j]);                                          \
                    }                                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                             \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
                            acc[i][j] = LOFT_MFMA_32x32x16(gf[i], xh[j], acc[i][j]);                          \
                            acc[i][j] = LOFT_MFMA_32x32x16(gf[i], xl[j], acc[i][j]);                          \
                        }                                                                                     \
                } while (0)
                if (RBM_ABL != 1) {
                    RBM_KSTEP(0);
                    if (kpad > 16) RBM_KSTEP(16);
                }
#undef RBM_KSTEP
            }
            // the next pair's tables ride behind this pair's MFMAs (other buffer; its readers finished two barriers ago)
            if (li + 1 < n) inv = tables(list[li + 1], buf ^ 1);
        }
    }
    }   // lists
    // ---- flush: lane holds pixel 32j + (lane & 31) and 4 x 4 consecutive channels per (i, gq)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = j * 32 + (lane & 31);
        const int y = ty0 + (pix >> 3), x = tx0 + (pix & 7);
        if (y >= H || x >= W) continue;
        bf16_t* gp = grad + (((size_t)b * H + y) * W + x) * C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = wave * 64 + i * 32 + 8 * gq + 4 * (lane >> 5);
                float v[4] = {acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]};
                if (accumulate) {
                    float o[4];
                    ld4(gp + n, o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += o[q];
                }
                st4(gp + n, v);
            }
    }
}

// ---- the same backward with the chunk as the unit of a software pipeline (LOFT_ROI_BWD_PIPE: built, bit-identical, NOT shipped) ----
// roi_align_bwd_mfma_kernel spends a pair (RoI, tile) like this: barrier, operand build, gout rows global -> registers -> LDS
// (an L2 / HBM round trip nothing of the workgroup overlaps), barrier, 8-16 MFMAs, the next pair's tables (two more dependent
// loads).  Its ablations (profiles/round2_probes/roi_bwd_ablations.txt) put 42 % of the launch in the gout staging, and a
// P = 14 RoI of a small building (bins below one pixel: 79 % of the mask list) lies in one or two tiles with up to 196 active
// bins = seven such chunks per pair.  Here
//   * gout rows of lists without rotations travel global -> LDS directly (1 KiB per wave access = two bins; the row swizzle is
//     applied on the global side: lane (row, physical chunk) fetches logical chunk q ^ ((row & 3) << 2)) into a DOUBLE-buffered
//     operand tile, and the copies of chunk c + 1 are issued right behind the barrier that publishes chunk c -- they are in
//     flight during chunk c's MFMAs, the tables of the pair after next and chunk c + 1's operand build; rotated (FOA) lists
//     keep the register path (four rows summed per bin) inside the same skeleton;
//   * the A operand is double-buffered too: ONE barrier per chunk instead of two to three;
//   * the sampling geometry of the pair after next is requested at the top of an iteration and consumed behind its barrier:
//     tables() waits for nothing, and "s_waitcnt vmcnt(0)" still means "the copies of the chunk about to be consumed landed";
//   * tables run two pairs ahead (three buffers): the pair whose first chunk is prefetched needs its bin masks one barrier early.
// Same arithmetic in the same order as the serial kernel: bit-identical maps (test_roi_align_bwd_pipe_bit_identical).
// Measured on the three lists of a bench step (tools/probes/roi_bwd_time.py, same box): 52 KiB of LDS = three workgroups per CU
// against the serial kernel's four -- mask list (2048 x 14^2) 506 -> 476 us, bbox list (8192 x 7^2) 551 -> 555, FOA list (rotated:
// no copies to pipeline, only the lost occupancy) 308 -> 353, the fused three-list launch 1084 -> 1106 us; with the geometry
// staged in LDS (59 KiB, two per CU) 1317 us.  Four resident workgroups hide the staging round trip as well as the pipeline
// does inside one, so the serial kernel stays the shipped one.
__global__ __launch_bounds__(256) void roi_align_bwd_mfma_pipe_kernel(RoiLevels L, int level, RoiBwdSets S,
                                                                      bf16_t* __restrict__ grad, int accumulate) {
    constexpr int C = 256;
    constexpr int GB = RBM_CHUNK * 512, AB = RBM_CHUNK * 256;
    __shared__ __attribute__((aligned(16))) char gbuf[2 * GB];               // 2 x [bin][256 ch] bf16, 16-byte chunks swizzled
    __shared__ __attribute__((aligned(16))) char abuf[2 * AB];               // 2 x [bin][64 px hi | 64 px lo] bf16
    __shared__ int list[RB_LIST];
    __shared__ int wcnt[4];
    __shared__ int range[2];
    __shared__ float WY[3][RB_MAXP][RB_TILE], WX[3][RB_MAXP][RB_TILE];
    __shared__ unsigned rnpt[16];
    __shared__ unsigned anyb[3][4];
    __shared__ float invb[3];
    const int H = L.H[level], W = L.W[level];
    const int tx0 = blockIdx.x * RB_TILE, ty0 = blockIdx.y * RB_TILE, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * GB / 16; i += 256) reinterpret_cast<uint4*>(gbuf)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 2 * AB / 16; i += 256) reinterpret_cast<uint4*>(abuf)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 16) rnpt[tid] = tid > 1 ? (65536u + (unsigned)tid - 1u) / (unsigned)tid : 65536u;
    const char* gfp[2];
    const char* afp[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) gfp[i] = roi_tr_frag_ptr<512>(gbuf, wave * 64 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        afp[j] = roi_tr_frag_ptr<256>(abuf, j * 32, lane);
        afp[2 + j] = roi_tr_frag_ptr<256>(abuf, 64 + j * 32, lane);
    }
    rf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bf16_t* gout = nullptr;
    const int4* rec = nullptr;
    int K = 0, P = 0, n_rot = 1;
    int c = 0;                               // chunks done so far: chunk c lives in operand buffers c & 1

    auto tables = [&](const float4 g0, const float4 g1, int buf) {     // a RoI's geometry -> WY / WX / anyb / invb [buf]
        RoiGeom g;
        g.start_h = g0.x; g.start_w = g0.y; g.bin_h = g0.z; g.bin_w = g0.w;
        g.grid_h = __float_as_int(g1.x); g.grid_w = __float_as_int(g1.y); g.count = g1.z;
        const int t = tid & 127;
        const int p = t >> 3, pix = t & 7;
        float w = 0.f;
        if (p < P) {
            if (tid < 128) {
                w = axis_weight(g.start_h, g.bin_h, g.grid_h, p, ty0 + pix, H);
                WY[buf][p][pix] = w;
            } else {
                w = axis_weight(g.start_w, g.bin_w, g.grid_w, p, tx0 + pix, W);
                WX[buf][p][pix] = w;
            }
        }
        unsigned long long bt = __ballot(w != 0.f);
        bt |= bt >> 4; bt |= bt >> 2; bt |= bt >> 1;
        bt &= 0x0101010101010101ull;
        if (lane == 0) anyb[buf][wave] = (unsigned)((bt * 0x0102040810204080ull) >> 56);
        if (tid == 0) invb[buf] = g1.w;
    };
    struct Pair { unsigned rmask, cmask, rnp; int npx, ka, kk; float inv; };
    auto pair_state = [&](int li) {          // (after a barrier behind tables(li))
        const int buf = li % 3;
        Pair q;
        q.rmask = anyb[buf][0] | (anyb[buf][1] << 8); q.cmask = anyb[buf][2] | (anyb[buf][3] << 8);
        q.npx = __popc(q.cmask);
        q.rnp = rnpt[q.npx];
        q.ka = __popc(q.rmask) * q.npx;
        q.kk = list[li];
        q.inv = invb[buf];
        return q;
    };
    // gout rows of chunk [kc, kc + kn) of pair q -> gb (lists without rotations): wave access op = bins 2 op, 2 op + 1
    auto issue_copies = [&](const Pair& q, int kc, int kn, char* gb) {
        const int nops = (kn + 1) >> 1;
        for (int op = wave; op < nops; op += 4) {
            const int kb = 2 * op + (lane >> 5);
            const int ab = kc + (kb < kn ? kb : kn - 1);                    // (an odd last access repeats the last bin: its A row is zero)
            const int qy = (int)(((unsigned)ab * q.rnp) >> 16);
            const int py = nth_set_bit(q.rmask, qy), px = nth_set_bit(q.cmask, ab - qy * q.npx);
            const int ql = (lane & 31) ^ ((kb & 3) << 2);
            const bf16_t* src = gout + ((size_t)q.kk * P * P + py * P + px) * C + ql * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(gb + op * 1024), 16, 0, 0);
        }
    };

#pragma unroll 1
    for (int si = 0; si < S.n; ++si) {
    gout = S.gout[si]; rec = S.rec[si];
    K = S.K[si]; P = S.P[si]; n_rot = S.n_rot[si];
    const bool dma = n_rot == 1;
    __syncthreads();
    if (tid < 2) {
        int lo = 0, hi = K;
        const int key = b + tid;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[mid].x < key) lo = mid + 1; else hi = mid; }
        range[tid] = lo;
    }
    __syncthreads();
    const int kbeg = S.sorted[si] ? range[0] : 0, kend = S.sorted[si] ? range[1] : K;
    for (int base = kbeg; base < kend; base += RB_LIST) {
        // ---- deterministic compaction of the RoIs that touch this tile
        const int k = base + tid;
        bool hit = false;
        if (k < kend) {
            const int4 r = rec[k];
            if (r.x == b && r.y == level) {
                const int x0 = r.z & 0xffff, x1 = r.z >> 16, y0 = r.w & 0xffff, y1 = r.w >> 16;
                hit = x1 >= tx0 && x0 < tx0 + RB_TILE && y1 >= ty0 && y0 < ty0 + RB_TILE;
            }
        }
        const unsigned long long bal = __ballot(hit);
        __syncthreads();                     // (previous batch: its last fragment and list reads are done)
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w2 = 0; w2 < wave; ++w2) off += wcnt[w2];
        const int n = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        if (n == 0) continue;
        const float4* geo = reinterpret_cast<const float4*>(rec + K);
        { const int k0 = list[0]; tables(geo[2 * k0], geo[2 * k0 + 1], 0); }
        if (n > 1) { const int k1 = list[1]; tables(geo[2 * k1], geo[2 * k1 + 1], 1); }
        __syncthreads();
        int li = 0, kc = 0;
        Pair cur = pair_state(0);
        if (dma && cur.ka > 0) issue_copies(cur, 0, min(RBM_CHUNK, cur.ka), gbuf + (c & 1) * GB);
#pragma unroll 1
        while (li < n) {
            const int kn = max(0, min(RBM_CHUNK, cur.ka - kc));
            const int kpad = (kn + 15) & ~15;
            char* ab_ = abuf + (c & 1) * AB;
            char* gb_ = gbuf + (c & 1) * GB;
            const int tb = li % 3;
            const bool same = kc + RBM_CHUNK < cur.ka;
            // geometry of the pair after next: requested here, used behind the barrier (tables() then waits for nothing, and the
            // s_waitcnt vmcnt(0) below -- which these two loads precede -- still means "this chunk's copies have landed")
            float4 gn0 = make_float4(0.f, 0.f, 0.f, 0.f), gn1 = gn0;
            if (!same && li + 2 < n) { const int k2 = list[li + 2]; gn0 = geo[2 * k2]; gn1 = geo[2 * k2 + 1]; }
            // ---- A[bin][pix] of chunk c
#pragma unroll 1
            for (int t = tid; t < kpad * 8; t += 256) {
                const int kb = t >> 3, y = t & 7;
                uint32_t hi4[4] = {0, 0, 0, 0}, lo4[4] = {0, 0, 0, 0};
                if (kb < kn) {
                    const int ab = kc + kb;
                    const int qy = (int)(((unsigned)ab * cur.rnp) >> 16);
                    const int py = nth_set_bit(cur.rmask, qy), px = nth_set_bit(cur.cmask, ab - qy * cur.npx);
                    const float wy = WY[tb][py][y] * cur.inv;
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const float a = wy * WX[tb][px][x];
                        const bf16_t h = f32_to_bf16(a);
                        const bf16_t l = f32_to_bf16(a - bf16_to_f32(h));
                        hi4[x >> 1] |= (uint32_t)h << (16 * (x & 1));
                        lo4[x >> 1] |= (uint32_t)l << (16 * (x & 1));
                    }
                }
                *reinterpret_cast<uint4*>(ab_ + kb * 256 + rwswz(kb, y) * 16) = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
                *reinterpret_cast<uint4*>(ab_ + kb * 256 + rwswz(kb, 8 + y) * 16) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
            }
            // ---- gout rows of chunk c: the copies issued one iteration ago, or (rotated lists) summed through registers now
            if (dma) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll 1
                for (int t = tid; t < kn * 32; t += 256) {
                    const int kb = t >> 5, q = t & 31;
                    const int ab = kc + kb;
                    const int qy = (int)(((unsigned)ab * cur.rnp) >> 16);
                    const int py = nth_set_bit(cur.rmask, qy), px = nth_set_bit(cur.cmask, ab - qy * cur.npx);
                    float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    uint4 rv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[r] = *reinterpret_cast<const uint4*>(gout + (((size_t)r * K + cur.kk) * P * P + rot_pos(py, px, P, r)) * C + q * 8);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float tv[8];
                        unpack8_16(rv[r], tv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sacc[e] += tv[e];
                    }
                    uint4 v;
                    v.x = (uint32_t)f32_to_bf16(sacc[0]) | ((uint32_t)f32_to_bf16(sacc[1]) << 16);
                    v.y = (uint32_t)f32_to_bf16(sacc[2]) | ((uint32_t)f32_to_bf16(sacc[3]) << 16);
                    v.z = (uint32_t)f32_to_bf16(sacc[4]) | ((uint32_t)f32_to_bf16(sacc[5]) << 16);
                    v.w = (uint32_t)f32_to_bf16(sacc[6]) | ((uint32_t)f32_to_bf16(sacc[7]) << 16);
                    *reinterpret_cast<uint4*>(gb_ + kb * 512 + rwswz(kb, q) * 16) = v;
                }
            }
            const int li2 = same ? li : li + 1, kc2 = same ? kc + RBM_CHUNK : 0;
            __syncthreads();                 // chunk c's operands are complete; chunk c - 1's fragment reads are done everywhere;
                                             // the tables written one or more iterations ago are visible
            Pair nxt = cur;
            if (!same && li2 < n) nxt = pair_state(li2);
            if (dma && li2 < n && nxt.ka - kc2 > 0)
                issue_copies(nxt, kc2, min(RBM_CHUNK, nxt.ka - kc2), gbuf + ((c + 1) & 1) * GB);
            if (kn > 0) {
                const int go = (c & 1) * GB, ao = (c & 1) * AB;
#define RBP_KSTEP(KS)                                                                                         \
                do {                                                                                          \
                    rbf16x8 gf[2], xh[2], xl[2];                                                              \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) gf[i] = roi_tr_frag_at<512, KS>(gfp[i] + go); \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
                        xh[j] = roi_tr_frag_at<256, KS>(afp[j] + ao);                                         \
                        xl[j] = roi_tr_frag_at<256, KS>(afp[2 + j] + ao);                                     \
                    }                                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                             \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
                            acc[i][j] = LOFT_MFMA_32x32x16(gf[i], xh[j], acc[i][j]);                          \
                            acc[i][j] = LOFT_MFMA_32x32x16(gf[i], xl[j], acc[i][j]);                          \
                        }                                                                                     \
                } while (0)
                RBP_KSTEP(0);
                if (kpad > 16) RBP_KSTEP(16);
#undef RBP_KSTEP
            }
            // the tables of the pair after next (their buffer's readers -- pair li - 1 -- finished before this iteration's barrier)
            if (!same && li + 2 < n) tables(gn0, gn1, (li + 2) % 3);
            cur = nxt; li = li2; kc = kc2; ++c;
        }
    }
    }   // lists
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = j * 32 + (lane & 31);
        const int y = ty0 + (pix >> 3), x = tx0 + (pix & 7);
        if (y >= H || x >= W) continue;
        bf16_t* gp = grad + (((size_t)b * H + y) * W + x) * C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = wave * 64 + i * 32 + 8 * gq + 4 * (lane >> 5);
                float v[4] = {acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]};
                if (accumulate) {
                    float o[4];
                    ld4(gp + n, o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += o[q];
                }
                st4(gp + n, v);
            }
    }
}

__global__ void roi_levels_kernel(const float* __restrict__ rois, int K, int num_levels, int finest_scale,
                                  int32_t* __restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = roi_level(rois + 5 * (size_t)k, num_levels, finest_scale);
}

static RoiLevels make_levels(const void* const* feats, const int* H, const int* W, const float* scales,
                             int num_levels, int finest_scale) {
    RoiLevels L;
    for (int i = 0; i < 4; ++i) {
        int j = i < num_levels ? i : num_levels - 1;
        L.feat[i] = feats ? feats[j] : nullptr;
        L.H[i] = H[j]; L.W[i] = W[j]; L.scale[i] = scales[j];
    }
    L.num_levels = num_levels;
    L.finest_scale = finest_scale;
    return L;
}

// ---- launch order of a RoI list (loft_roi_order) ---------------------------------------------------------------------------
// One workgroup: counting sort of the K RoIs by (image, pyramid level, row strip of that level's map).  The order INSIDE a bucket
// is whatever the LDS atomics give -- it only decides which workgroup computes which RoI, every RoI's arithmetic and output row
// are untouched, so results are bit-identical to the unordered launch.
#define RO_BINS 2048
__global__ __launch_bounds__(1024) void roi_order_kernel(RoiLevels L, const float* __restrict__ rois, int K, int B,
                                                         int32_t* __restrict__ order) {
    __shared__ int hist[RO_BINS];
    __shared__ int wsum[16];
    const int tid = threadIdx.x;
    int nstrip = RO_BINS / (4 * (B > 0 ? B : 1));
    nstrip = nstrip > 32 ? 32 : (nstrip < 1 ? 1 : nstrip);
    auto key_of = [&](int k) {
        const float* r = rois + 5 * (size_t)k;
        int b = (int)r[0];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);
        const int lv = L.num_levels > 1 ? roi_level(r, L.num_levels, L.finest_scale) : 0;
        const float cy = 0.5f * (r[2] + r[4]) * L.scale[lv];
        int st = (int)(cy * (float)nstrip / (float)L.H[lv]);
        st = st < 0 ? 0 : (st >= nstrip ? nstrip - 1 : st);
        const int key = (b * 4 + lv) * nstrip + st;
        return key < RO_BINS ? key : RO_BINS - 1;
    };
    for (int i = tid; i < RO_BINS; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += 1024) atomicAdd(&hist[key_of(k)], 1);
    __syncthreads();
    // exclusive scan of the 2048 bins: two per thread, wave scan, then the 16 wave totals
    const int a = hist[2 * tid], c = hist[2 * tid + 1];
    int v = a + c;
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if ((tid & 63) >= d) v += u;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = v;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    const int excl = base + v - (a + c);
    __syncthreads();
    hist[2 * tid] = excl;
    hist[2 * tid + 1] = excl + a;
    __syncthreads();
    for (int k = tid; k < K; k += 1024) order[atomicAdd(&hist[key_of(k)], 1)] = k;
}

LOFT_EXPORT int loft_roi_order(const int* H, const float* scales, int num_levels, int finest_scale, const float* rois, int K,
                               int B, int32_t* order, void* stream) {
    if (K <= 0) return 0;
    if (num_levels < 1 || num_levels > 4 || B < 1) return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(nullptr, H, H, scales, num_levels, finest_scale);
    hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, L, rois, K, B, order);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_roi_align_fwd_ord(const void* const* feats, const int* H, const int* W, const float* scales,
                                       int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                       int P, int n_rot, void* out, int variant, const int32_t* order, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if (K <= 0) return 0;
    if (num_levels < 1 || num_levels > 4 || (C & 3) || (n_rot != 1 && n_rot != 4)) return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(feats, H, W, scales, num_levels, finest_scale);
    hipStream_t s = (hipStream_t)stream;
    const int kern = variant & 0xff;                 // bits 8-17: tuning of the 16-byte separable kernel (below)
    if ((kern != LOFT_ROI_AUTO && kern != LOFT_ROI_FWD_SAMPLE && kern != LOFT_ROI_FWD_SEP4) || (variant & ~0x1ffffff) ||
        (kern != LOFT_ROI_AUTO && (variant >> 8)))
        return (int)hipErrorInvalidValue;
    const bool sample_form = kern == LOFT_ROI_FWD_SAMPLE;         // the sample-order kernel also for the 16-bit type (tests, A/B)
    const dim3 grid(order ? 8 * ((K + 7) / 8) : K);
    if (dtype == LOFT_ACT16 && !sample_form && !(C & 7) && kern != LOFT_ROI_FWD_SEP4)
    {
        // Tuning bits of `variant` (tools/probes/roi_fwd_time.py; 0 everywhere = the shipped choice):
        //   8-15  stage size of the LDS form in KiB (shipped 24: 632 us per step for the three lists of a bench step against 664
        //         at 40 and 672 with none; 255 = none -- every RoI streams or samples)
        //   16-17 smallest channel block of the LDS form = 64 << bits
        //   18-21 most samples per bin for which a RoI whose bins are a pixel or more still takes the LDS form (shipped 4)
        //   22-24 workgroups per RoI (shipped: 2 for lists of up to 4096 RoIs -- eight workgroups per CU of very uneven length
        //         leave the CUs 1.8 deep of 3-4 possible; 1 beyond: the duplicated set-up costs more than the tail there)
        const int kb = (variant >> 8) & 0xff, stage_bytes = kb == 255 ? 0 : (kb ? kb : 24) * 1024, min_cb = 64 << ((variant >> 16) & 3);
        const int gm = (variant >> 18) & 15, gmax = gm ? gm : 4;
        const int sp = (variant >> 22) & 7, nsplit = sp ? sp : (K <= 4096 ? 2 : 1);
        if (stage_bytes > 48 * 1024)
            hipFuncSetAttribute((const void*)roi_align_fwd_sep8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL(roi_align_fwd_sep8_kernel, dim3(grid.x * nsplit), dim3(256), stage_bytes, s, L, rois, K, C, P, n_rot, (bf16_t*)out, order,
                           stage_bytes, min_cb, gmax, nsplit);
    }
    else if (dtype == LOFT_ACT16 && !sample_form)
        hipLaunchKernelGGL(roi_align_fwd_sep_kernel, grid, dim3(256), 0, s, L, rois, K, C, P, n_rot, (bf16_t*)out, order);
    else if (dtype == LOFT_ACT16)
        hipLaunchKernelGGL(roi_align_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, L, rois, K, C, P, n_rot, (bf16_t*)out, order);
    else if (dtype == LOFT_F32)
        hipLaunchKernelGGL(roi_align_fwd_kernel<float>, grid, dim3(256), 0, s, L, rois, K, C, P, n_rot, (float*)out, order);
    else
        return (int)hipErrorInvalidValue;
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_roi_align_fwd_v(const void* const* feats, const int* H, const int* W, const float* scales,
                                     int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                     int P, int n_rot, void* out, int variant, void* stream) {
    return loft_roi_align_fwd_ord(feats, H, W, scales, num_levels, finest_scale, C, dtype, rois, K, P, n_rot, out, variant, nullptr,
                                  stream);
}

LOFT_EXPORT int loft_roi_align_fwd(const void* const* feats, const int* H, const int* W, const float* scales,
                                   int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                   int P, int n_rot, void* out, void* stream) {
    return loft_roi_align_fwd_v(feats, H, W, scales, num_levels, finest_scale, C, dtype, rois, K, P, n_rot, out, LOFT_ROI_AUTO, stream);
}

LOFT_EXPORT int loft_roi_align_bwd_v(void* const* grad_feats, const int* H, const int* W, const float* scales,
                                     int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                     int P, int n_rot, const void* grad_out, int B, int accumulate, int rois_sorted,
                                     void* workspace, int grad_dtype, int variant, void* stream) {
    if (variant != LOFT_ROI_AUTO && variant != LOFT_ROI_BWD_VALU && variant != LOFT_ROI_BWD_PIPE) return (int)hipErrorInvalidValue;
    const bool valu_form = variant == LOFT_ROI_BWD_VALU;          // the tile-owner VALU kernel instead of the per-pair GEMMs
    const bool serial_form = variant != LOFT_ROI_BWD_PIPE;        // (shipped) the per-pair GEMMs without the chunk pipeline
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if (num_levels < 1 || num_levels > 4 || (C & 3) || (n_rot != 1 && n_rot != 4) || P > RB_MAXP)
        return (int)hipErrorInvalidValue;
    if (grad_dtype != LOFT_F32 && !(grad_dtype == LOFT_ACT16 && dtype == LOFT_ACT16)) return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(nullptr, H, W, scales, num_levels, finest_scale);
    hipStream_t s = (hipStream_t)stream;
    int4* rec = (int4*)workspace;
    if (K > 0) {
        hipLaunchKernelGGL(roi_prep_kernel, dim3(loft_cdiv(K, 256)), dim3(256), 0, s, L, rois, K, P, rec);
        LOFT_LAUNCH_CHECK();
    }
    RoiBwdSets one = {};
    one.rois[0] = rois; one.gout[0] = (const bf16_t*)grad_out; one.rec[0] = rec;
    one.K[0] = K; one.P[0] = P; one.n_rot[0] = n_rot; one.sorted[0] = rois_sorted; one.n = 1;
    for (int l = 0; l < num_levels; ++l) {
        dim3 grid(loft_cdiv(W[l], RB_TILE), loft_cdiv(H[l], RB_TILE), B);
        if (dtype == LOFT_ACT16 && grad_dtype == LOFT_ACT16 && C == 256 && !valu_form && serial_form)
            hipLaunchKernelGGL(roi_align_bwd_mfma_kernel, grid, dim3(256), 0, s, L, l, one, (bf16_t*)grad_feats[l], accumulate);
        else if (dtype == LOFT_ACT16 && grad_dtype == LOFT_ACT16 && C == 256 && !valu_form)
            hipLaunchKernelGGL(roi_align_bwd_mfma_pipe_kernel, grid, dim3(256), 0, s, L, l, one, (bf16_t*)grad_feats[l], accumulate);
        else if (dtype == LOFT_ACT16 && grad_dtype == LOFT_ACT16)
            hipLaunchKernelGGL((roi_align_bwd_tile_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, L, l, rois, K, C, P, n_rot,
                               (const bf16_t*)grad_out, (bf16_t*)grad_feats[l], accumulate, rec, rois_sorted);
        else if (dtype == LOFT_ACT16)
            hipLaunchKernelGGL((roi_align_bwd_tile_kernel<bf16_t, float>), grid, dim3(256), 0, s, L, l, rois, K, C, P, n_rot,
                               (const bf16_t*)grad_out, (float*)grad_feats[l], accumulate, rec, rois_sorted);
        else if (dtype == LOFT_F32)
            hipLaunchKernelGGL((roi_align_bwd_tile_kernel<float, float>), grid, dim3(256), 0, s, L, l, rois, K, C, P, n_rot,
                               (const float*)grad_out, (float*)grad_feats[l], accumulate, rec, rois_sorted);
        else
            return (int)hipErrorInvalidValue;
        LOFT_LAUNCH_CHECK();
    }
    return 0;
}

// Several RoI lists over the same pyramid in one pass per level (16-bit maps, C = 256: the LOFT head's three extractors).
// Every list brings its own rois / K / P / n_rot / grad_out / sorted flag / 48*K-byte workspace; other configurations
// run list after list through loft_roi_align_bwd_v (the later ones accumulating).
LOFT_EXPORT int loft_roi_align_bwd_multi(void* const* grad_feats, const int* H, const int* W, const float* scales,
                                         int num_levels, int finest_scale, int C, int dtype, int nsets,
                                         const float* const* rois, const int* K, const int* P, const int* n_rot,
                                         const void* const* grad_out, int B, int accumulate, const int* rois_sorted,
                                         void* const* workspace, int grad_dtype, void* stream) {
    if (nsets < 1) return (int)hipErrorInvalidValue;
    const bool fused = dtype == LOFT_ACT16 && grad_dtype == LOFT_ACT16 && C == 256 && nsets <= RBM_SETS;
    if (!fused) {
        for (int i = 0; i < nsets; ++i) {
            int rc = loft_roi_align_bwd_v(grad_feats, H, W, scales, num_levels, finest_scale, C, dtype, rois[i], K[i], P[i],
                                          n_rot[i], grad_out[i], B, accumulate || i > 0, rois_sorted[i], workspace[i],
                                          grad_dtype, LOFT_ROI_AUTO, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (num_levels < 1 || num_levels > 4) return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(nullptr, H, W, scales, num_levels, finest_scale);
    hipStream_t s = (hipStream_t)stream;
    RoiBwdSets S = {};
    for (int i = 0; i < nsets; ++i) {
        if ((n_rot[i] != 1 && n_rot[i] != 4) || P[i] > RB_MAXP || K[i] < 0) return (int)hipErrorInvalidValue;
        if (K[i] == 0) continue;
        const int j = S.n++;
        S.rois[j] = rois[i]; S.gout[j] = (const bf16_t*)grad_out[i]; S.rec[j] = (const int4*)workspace[i];
        S.K[j] = K[i]; S.P[j] = P[i]; S.n_rot[j] = n_rot[i]; S.sorted[j] = rois_sorted[i];
        hipLaunchKernelGGL(roi_prep_kernel, dim3(loft_cdiv(K[i], 256)), dim3(256), 0, s, L, rois[i], K[i], P[i],
                           (int4*)workspace[i]);
        LOFT_LAUNCH_CHECK();
    }
    for (int l = 0; l < num_levels; ++l) {
        dim3 grid(loft_cdiv(W[l], RB_TILE), loft_cdiv(H[l], RB_TILE), B);
#ifdef RBM_PIPE_MULTI        /* A/B builds: the chunk-pipelined kernel on the multi-list route */
        hipLaunchKernelGGL(roi_align_bwd_mfma_pipe_kernel, grid, dim3(256), 0, s, L, l, S, (bf16_t*)grad_feats[l], accumulate);
#else
        hipLaunchKernelGGL(roi_align_bwd_mfma_kernel, grid, dim3(256), 0, s, L, l, S, (bf16_t*)grad_feats[l], accumulate);
#endif
        LOFT_LAUNCH_CHECK();
    }
    return 0;
}

LOFT_EXPORT int loft_roi_align_bwd(void* const* grad_feats, const int* H, const int* W, const float* scales,
                                   int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                   int P, int n_rot, const void* grad_out, int B, int accumulate, int rois_sorted,
                                   void* workspace, int grad_dtype, void* stream) {
    return loft_roi_align_bwd_v(grad_feats, H, W, scales, num_levels, finest_scale, C, dtype, rois, K, P, n_rot, grad_out, B,
                                accumulate, rois_sorted, workspace, grad_dtype, LOFT_ROI_AUTO, stream);
}

LOFT_EXPORT int loft_map_roi_levels(const float* rois, int K, int num_levels, int finest_scale, int32_t* out,
                                    void* stream) {
    if (K <= 0) return 0;
    hipLaunchKernelGGL(roi_levels_kernel, dim3(loft_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, rois, K,
                       num_levels, finest_scale, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}
