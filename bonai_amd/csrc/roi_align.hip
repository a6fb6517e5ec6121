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

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(RoiLevels L, const float* __restrict__ rois, int K, int C,
                                                            int P, int n_rot, T* __restrict__ out) {
    const int k = blockIdx.x;
    if (k >= K) return;
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

// ---- backward: tile-owner gather, no global atomics -------------------------------------------
// One workgroup OWNS an 8x8-pixel tile of one image's gradient map at one pyramid level and keeps it
// in LDS as fp32 [64 px][C] (64 KiB at C=256).  It scans the RoI list, compacts the RoIs of its
// (image, level) whose footprint touches the tile, and for each of them replays the bins / bilinear
// samples whose corners land inside the tile with LDS float atomics (ds_add_f32).  The tile is then
// written (or added, when several extractors feed the same map) to HBM exactly once, coalesced.
// Versus the scatter formulation (one global fp32 atomic per sample-corner-channel, ~3.7e9 per step at
// batch 8) this moves all accumulation traffic into LDS; HBM sees each gradient pixel once.
#define RB_TILE 8
#define RB_LIST 256

// prepass: one record per RoI = (batch, level, clamped pixel footprint) so the tile owners scan 16-byte
// records instead of redoing sqrt/log2 per (tile, RoI) pair.
__global__ void roi_prep_kernel(RoiLevels L, const float* __restrict__ rois, int K, int P, int4* __restrict__ rec) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* roi = rois + 5 * (size_t)k;
    const RoiGeom g = roi_geom(roi, L, P);
    const int H = L.H[g.level], W = L.W[g.level];
    const float end_w = g.start_w + g.bin_w * (float)P, end_h = g.start_h + g.bin_h * (float)P;
    const float lo_x = fminf(g.start_w, end_w) - 1.f, hi_x = fmaxf(g.start_w, end_w) + 1.f;
    const float lo_y = fminf(g.start_h, end_h) - 1.f, hi_y = fmaxf(g.start_h, end_h) + 1.f;
    const int x0 = (int)floorf(fminf(fmaxf(lo_x, 0.f), (float)(W - 1))), x1 = (int)ceilf(fminf(fmaxf(hi_x, 0.f), (float)(W - 1)));
    const int y0 = (int)floorf(fminf(fmaxf(lo_y, 0.f), (float)(H - 1))), y1 = (int)ceilf(fminf(fmaxf(hi_y, 0.f), (float)(H - 1)));
    rec[k] = make_int4(g.batch, g.level, x0 | (x1 << 16), y0 | (y1 << 16));
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_tile_kernel(RoiLevels L, int level, const float* __restrict__ rois,
                                                                 int K, int C, int P, int n_rot,
                                                                 const T* __restrict__ gout, float* __restrict__ grad,
                                                                 int accumulate, const int4* __restrict__ rec, int sorted) {
    extern __shared__ __attribute__((aligned(16))) float acc[];  // [64][C]
    __shared__ int list[RB_LIST];
    __shared__ int nlist;
    const int H = L.H[level], W = L.W[level];
    const int tx0 = blockIdx.x * RB_TILE, ty0 = blockIdx.y * RB_TILE, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = C >> 2;
    for (int i = tid; i < RB_TILE * RB_TILE * C; i += 256) acc[i] = 0.f;
    // RoIs are image-major (bbox2roi order): binary-search this image's range once, then scan only it
    __shared__ int range[2];
    if (tid < 2) {
        int lo = 0, hi = K;
        const int key = b + tid;   // first record with batch >= b (tid 0) / >= b+1 (tid 1)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[mid].x < key) lo = mid + 1; else hi = mid; }
        range[tid] = lo;
    }
    __syncthreads();
    const int kbeg = sorted ? range[0] : 0, kend = sorted ? range[1] : K;
    for (int base = kbeg; base < kend; base += RB_LIST) {  // chunks that always fit the list
        if (tid == 0) nlist = 0;
        __syncthreads();
        const int k = base + tid;
        if (k < kend) {
            const int4 r = rec[k];
            if (r.x == b && r.y == level) {
                const int x0 = r.z & 0xffff, x1 = r.z >> 16, y0 = r.w & 0xffff, y1 = r.w >> 16;
                if (x1 >= tx0 && x0 < tx0 + RB_TILE && y1 >= ty0 && y0 < ty0 + RB_TILE) list[atomicAdd(&nlist, 1)] = k;
            }
        }
        __syncthreads();
        const int n = nlist;
        for (int li = 0; li < n; ++li) {
            const int kk = list[li];
            const RoiGeom g = roi_geom(rois + 5 * (size_t)kk, L, P);
            for (int bin = wave; bin < P * P; bin += 4) {
                const int py = bin / P, px = bin - py * P;
                // cheap reject: the bin's sample span (+1 px for the high corner) vs the tile, in clamped coordinates
                const float by0 = g.start_h + py * g.bin_h, by1 = by0 + g.bin_h;
                const float bx0 = g.start_w + px * g.bin_w, bx1 = bx0 + g.bin_w;
                const float qy0 = fminf(fmaxf(fminf(by0, by1) - 1.f, 0.f), (float)(H - 1));
                const float qy1 = fminf(fmaxf(fmaxf(by0, by1) + 1.f, 0.f), (float)(H - 1));
                const float qx0 = fminf(fmaxf(fminf(bx0, bx1) - 1.f, 0.f), (float)(W - 1));
                const float qx1 = fminf(fmaxf(fmaxf(bx0, bx1) + 1.f, 0.f), (float)(W - 1));
                if (qy1 < (float)ty0 || qy0 >= (float)(ty0 + RB_TILE) || qx1 < (float)tx0 || qx0 >= (float)(tx0 + RB_TILE))
                    continue;
                for (int c4 = lane; c4 < cg; c4 += 64) {
                    const int c0 = c4 << 2;
                    float gv[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int r = 0; r < n_rot; ++r) {
                        float t[4];
                        ld4(gout + (((size_t)r * K + kk) * P * P + rot_pos(py, px, P, r)) * C + c0, t);
#pragma unroll
                        for (int q = 0; q < 4; ++q) gv[q] += t[q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) gv[q] /= g.count;
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        const float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            const float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                            const Bil bl = bil_setup(y, x, H, W);
                            if (!bl.valid) continue;
                            const int ys[2] = {bl.y_low - ty0, bl.y_high - ty0};
                            const int xs[2] = {bl.x_low - tx0, bl.x_high - tx0};
                            const float ws[4] = {bl.w1, bl.w2, bl.w3, bl.w4};
#pragma unroll
                            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                                for (int cx = 0; cx < 2; ++cx) {
                                    if ((unsigned)ys[cy] < RB_TILE && (unsigned)xs[cx] < RB_TILE) {
                                        float* a = acc + (size_t)(ys[cy] * RB_TILE + xs[cx]) * C + c0;
                                        const float wgt = ws[cy * 2 + cx];
#pragma unroll
                                        for (int q = 0; q < 4; ++q) atomicAdd(a + q, gv[q] * wgt);
                                    }
                                }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // flush the tile, one coalesced pass
    for (int i = tid; i < RB_TILE * RB_TILE * cg; i += 256) {
        const int pix = i / cg, c0 = (i - pix * cg) << 2;
        const int y = ty0 + pix / RB_TILE, x = tx0 + pix % RB_TILE;
        if (y >= H || x >= W) continue;
        float* gp = grad + (((size_t)b * H + y) * W + x) * C + c0;
        float v[4] = {acc[pix * C + c0], acc[pix * C + c0 + 1], acc[pix * C + c0 + 2], acc[pix * C + c0 + 3]};
        if (accumulate) {
            float o[4];
            ld4(gp, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += o[q];
        }
        st4(gp, v);
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

LOFT_EXPORT int loft_roi_align_fwd(const void* const* feats, const int* H, const int* W, const float* scales,
                                   int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                   int P, int n_rot, void* out, void* stream) {
    if (K <= 0) return 0;
    if (num_levels < 1 || num_levels > 4 || (C & 3) || (n_rot != 1 && n_rot != 4)) return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(feats, H, W, scales, num_levels, finest_scale);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LOFT_BF16)
        hipLaunchKernelGGL(roi_align_fwd_kernel<bf16_t>, dim3(K), dim3(256), 0, s, L, rois, K, C, P, n_rot, (bf16_t*)out);
    else if (dtype == LOFT_F32)
        hipLaunchKernelGGL(roi_align_fwd_kernel<float>, dim3(K), dim3(256), 0, s, L, rois, K, C, P, n_rot, (float*)out);
    else
        return (int)hipErrorInvalidValue;
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_roi_align_bwd(float* const* grad_feats, const int* H, const int* W, const float* scales,
                                   int num_levels, int finest_scale, int C, int dtype, const float* rois, int K,
                                   int P, int n_rot, const void* grad_out, int B, int accumulate, int rois_sorted,
                                   void* workspace, void* stream) {
    if (num_levels < 1 || num_levels > 4 || (C & 3) || (n_rot != 1 && n_rot != 4) || C > 512)
        return (int)hipErrorInvalidValue;
    RoiLevels L = make_levels(nullptr, H, W, scales, num_levels, finest_scale);
    hipStream_t s = (hipStream_t)stream;
    int4* rec = (int4*)workspace;
    if (K > 0) {
        hipLaunchKernelGGL(roi_prep_kernel, dim3(loft_cdiv(K, 256)), dim3(256), 0, s, L, rois, K, P, rec);
        LOFT_LAUNCH_CHECK();
    }
    const size_t sh = (size_t)RB_TILE * RB_TILE * C * sizeof(float);
    for (int l = 0; l < num_levels; ++l) {
        dim3 grid(loft_cdiv(W[l], RB_TILE), loft_cdiv(H[l], RB_TILE), B);
        if (sh > 60000) {  // 64 KiB tile + the static list exceeds the default 64 KiB dynamic-LDS cap
            hipError_t e = dtype == LOFT_BF16
                ? hipFuncSetAttribute((const void*)roi_align_bwd_tile_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)
                : hipFuncSetAttribute((const void*)roi_align_bwd_tile_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (e != hipSuccess) return (int)e;
        }
        if (dtype == LOFT_BF16)
            hipLaunchKernelGGL(roi_align_bwd_tile_kernel<bf16_t>, grid, dim3(256), sh, s, L, l, rois, K, C, P, n_rot,
                               (const bf16_t*)grad_out, grad_feats[l], accumulate, rec, rois_sorted);
        else if (dtype == LOFT_F32)
            hipLaunchKernelGGL(roi_align_bwd_tile_kernel<float>, grid, dim3(256), sh, s, L, l, rois, K, C, P, n_rot,
                               (const float*)grad_out, grad_feats[l], accumulate, rec, rois_sorted);
        else
            return (int)hipErrorInvalidValue;
        LOFT_LAUNCH_CHECK();
    }
    return 0;
}

LOFT_EXPORT int loft_map_roi_levels(const float* rois, int K, int num_levels, int finest_scale, int32_t* out,
                                    void* stream) {
    if (K <= 0) return 0;
    hipLaunchKernelGGL(roi_levels_kernel, dim3(loft_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, rois, K,
                       num_levels, finest_scale, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}
