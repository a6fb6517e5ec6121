// deform.hip -- modulated deformable convolution (DCNv2) sampling for gfx950, NHWC.
//
// Reference call sites: mmdet/models/backbones/resnet.py:171-194 (Bottleneck.conv2 built from dcn=dict(type='DCNv2')),
// :608-612 (conv_offset zero init), mmdet/models/necks/fpn.py:116-132 (conv_cfg=dict(type='DCNv2')).  The arithmetic
// lives in mmcv==1.0.5 `ModulatedDeformConv2dPack` / `modulated_deform_conv2d` (not in /root/reference); this file follows
// its published algorithm (Zhu et al., "Deformable ConvNets v2", and the modulated_deformable_im2col definition):
//
//   raw  = conv_offset(x)                          [B, 3*DG*K, OH, OW]  (K = kh*kw, DG = deform groups)
//   off  = raw[:, 0 : 2*DG*K]                      channel g*2K + 2k = dy, g*2K + 2k + 1 = dx of tap k, group g
//   mask = sigmoid(raw[:, 2*DG*K + g*K + k])
//   col[m, k, c] = mask * bilinear0(x[b, :, :, c], oy*s - p + i*d + dy, ox*s - p + j*d + dx)     (k = i*kw + j)
//   bilinear0: zero unless -1 < y < H and -1 < x < W; corners outside the map contribute zero.
//   y = col (as a [M, K*C] matrix) x W[Cout, K*C]^T  -- run by loft_conv_tap_* as a 1x1 contraction.
//
// Layout / mapping: one workgroup owns PIX consecutive output pixels.  Phase 1 stages their offsets and mask logits through
// LDS and turns every (pixel, tap) into an "item": 4 corner element offsets (or -1) + 4 bilinear weights + the mask.
// Phase 2 sweeps items x 8-channel groups with consecutive lanes on consecutive channel groups, so each corner read and
// each col write is a 16-byte (bf16) / 32-byte (fp32) access contiguous across the lanes of an item.
// Roofline: HBM bandwidth.  Algorithmic bytes per output pixel: K*C*sizeof(T) written + <= 4*K*C*sizeof(T) read (L2 absorbs
// most corner re-reads: neighbouring taps/pixels hit the same rows) + 3*DG*K*4 offset bytes.
//
// Backward: given dcol, produce d(raw) (offsets and mask logits, including the sigmoid derivative) and
// d(x) += mask * w_corner * dcol.  Three implementations, selected by loft_mdcn_sample_bwd:
//   * binned   (bf16, deform_groups == 1, C % 64 == 0: the training path) -- per output tile the contributions are counting-
//     sorted in LDS by the window pixel they land on, then accumulated in registers per pixel; no float atomics at all.
//   * window   (fp32 parity / test path) -- lane = channel, fp32 LDS window with wave-private rows.
//   * generic  (any deform_groups / C % 8 == 0) -- global fp32 atomics (~2.3 G transactions/s on MI355X: slow, kept for coverage).
// Both tiled kernels store per-tile gradient windows to a workspace; mdcn_window_gather_kernel sums the overlaps.
// Measured (tools/bench_mdcn.py, 8 x 256 x 256^2 bf16 3x3): generic 15.7 ms avg per launch -> binned 4.6 ms at the largest
// shape (fwd 1.3 ms).
#include "loft_common.h"
#include "../../include/loft_hip.h"
#include <stdlib.h>

namespace {

constexpr int MDCN_PIX = 8;       // output pixels per workgroup
constexpr int MDCN_MAXK = 9;      // taps (3x3)
constexpr int MDCN_MAXDG = 4;
constexpr int MDCN_ITEMS = MDCN_PIX * MDCN_MAXK * MDCN_MAXDG;

struct MdcnArgs {
    int B, IH, IW, C, OH, OW, kh, kw, stride, pad, dil, DG, omc;
    long M;
};

struct Item {
    int off[4];      // element offset of the corner pixel's channel 0 (-1: outside)
    float w[4];      // hh*hw, hh*lw, lh*hw, lh*lw
    float hh, hw, lh, lw, mask;
    int inside;
};

__device__ __forceinline__ void ld8v(const bf16_t* p, float v[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack8_16(t, v);
}
__device__ __forceinline__ void ld8v(const float* p, float v[8]) { ld4(p, v); ld4(p + 4, v + 4); }
__device__ __forceinline__ void st8v(bf16_t* p, const float v[8]) { st4(p, v); st4(p + 4, v + 4); }
__device__ __forceinline__ void st8v(float* p, const float v[8]) { st4(p, v); st4(p + 4, v + 4); }

// Phase 1 (shared by forward and backward): build the items of this workgroup's pixels.
__device__ __forceinline__ void build_items(const MdcnArgs& a, const float* __restrict__ om, long m0, int npix, Item* items,
                                            float* raw) {
    const int K = a.kh * a.kw, KG = K * a.DG;
    const int nraw = 3 * KG;
    // stage the raw conv_offset outputs of npix pixels: [npix][3*DG*K] floats, coalesced over channels
    for (int i = threadIdx.x; i < npix * nraw; i += blockDim.x) {
        const int p = i / nraw, ch = i - p * nraw;
        raw[i] = om[(m0 + p) * a.omc + ch];
    }
    __syncthreads();
    const int ohw = a.OH * a.OW;
    for (int i = threadIdx.x; i < npix * KG; i += blockDim.x) {
        const int p = i / KG, r = i - p * KG;
        const int g = r / K, k = r - g * K;
        const int ki = k / a.kw, kj = k - ki * a.kw;
        const long m = m0 + p;
        const int b = (int)(m / ohw);
        const int rem = (int)(m - (long)b * ohw);
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        const float* rp = raw + p * nraw;
        const float dy = rp[g * 2 * K + 2 * k], dx = rp[g * 2 * K + 2 * k + 1];
        const float ml = rp[2 * KG + g * K + k];
        const float h = (float)(oy * a.stride - a.pad + ki * a.dil) + dy;
        const float w = (float)(ox * a.stride - a.pad + kj * a.dil) + dx;
        Item it;
        it.mask = 1.f / (1.f + expf(-ml));
        it.inside = (h > -1.f) && (w > -1.f) && (h < (float)a.IH) && (w < (float)a.IW);
        const float hf = floorf(h), wf = floorf(w);
        const int hl = (int)hf, wl = (int)wf, hh_i = hl + 1, wh_i = wl + 1;
        it.lh = h - hf; it.lw = w - wf; it.hh = 1.f - it.lh; it.hw = 1.f - it.lw;
        it.w[0] = it.hh * it.hw; it.w[1] = it.hh * it.lw; it.w[2] = it.lh * it.hw; it.w[3] = it.lh * it.lw;
        const bool t0 = hl >= 0, t1 = hh_i <= a.IH - 1, l0 = wl >= 0, l1 = wh_i <= a.IW - 1;
        const long base = (long)b * a.IH * a.IW;
        const bool in = it.inside;
        it.off[0] = (in && t0 && l0) ? (int)((base + (long)hl * a.IW + wl)) : -1;
        it.off[1] = (in && t0 && l1) ? (int)((base + (long)hl * a.IW + wh_i)) : -1;
        it.off[2] = (in && t1 && l0) ? (int)((base + (long)hh_i * a.IW + wl)) : -1;
        it.off[3] = (in && t1 && l1) ? (int)((base + (long)hh_i * a.IW + wh_i)) : -1;
        items[i] = it;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void mdcn_sample_fwd_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                              T* __restrict__ col, const MdcnArgs a) {
    __shared__ Item items[MDCN_ITEMS];
    __shared__ float raw[MDCN_PIX * 3 * MDCN_MAXK * MDCN_MAXDG];
    const long m0 = (long)blockIdx.x * MDCN_PIX;
    const int npix = (int)((a.M - m0) < MDCN_PIX ? (a.M - m0) : MDCN_PIX);
    build_items(a, om, m0, npix, items, raw);
    const int K = a.kh * a.kw, KG = K * a.DG;
    const int cg = a.C >> 3, cpg = cg / a.DG;          // 8-channel groups per pixel / per deformable group
    const int total = npix * K * cg;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c8 = i % cg;
        const int pk = i / cg;                           // p*K + k
        const int p = pk / K, k = pk - p * K;
        const int g = c8 / cpg;
        const Item& it = items[p * KG + g * K + k];
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
            const int o = it.off[cnr];
            if (o >= 0) {
                float v[8];
                ld8v(x + (long)o * a.C + c8 * 8, v);
                const float wq = it.w[cnr];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = acc[q] + wq * v[q];
            }
        }
        const float mk = it.mask;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = acc[q] * mk;
        st8v(col + ((m0 + p) * K + k) * a.C + c8 * 8, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mdcn_sample_bwd_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                              const T* __restrict__ dcol, float* __restrict__ dx,
                                                              float* __restrict__ dom, const MdcnArgs a) {
    __shared__ Item items[MDCN_ITEMS];
    __shared__ float raw[MDCN_PIX * 3 * MDCN_MAXK * MDCN_MAXDG];
    __shared__ float red[MDCN_ITEMS][3];                // d(dy), d(dx), d(mask) per item
    const long m0 = (long)blockIdx.x * MDCN_PIX;
    const int npix = (int)((a.M - m0) < MDCN_PIX ? (a.M - m0) : MDCN_PIX);
    for (int i = threadIdx.x; i < MDCN_ITEMS * 3; i += blockDim.x) (&red[0][0])[i] = 0.f;
    build_items(a, om, m0, npix, items, raw);
    const int K = a.kh * a.kw, KG = K * a.DG;
    const int cg = a.C >> 3, cpg = cg / a.DG;
    const int seg = cpg < 64 ? cpg : 64;                // lanes of one item inside a wave (power of two, C % (8*DG) == 0)
    const int total = npix * K * cg;
    const int rounds = (total + blockDim.x - 1) / blockDim.x;
    for (int r = 0; r < rounds; ++r) {
        const int i = r * blockDim.x + threadIdx.x;
        const bool live = i < total;
        float sdy = 0.f, sdx = 0.f, sdm = 0.f;
        int item_idx = 0;
        if (live) {
            const int c8 = i % cg;
            const int pk = i / cg;
            const int p = pk / K, k = pk - p * K;
            const int g = c8 / cpg;
            item_idx = p * KG + g * K + k;
            const Item& it = items[item_idx];
            if (it.inside) {
                float d[8], v[4][8];
                ld8v(dcol + ((m0 + p) * K + k) * a.C + c8 * 8, d);
#pragma unroll
                for (int cnr = 0; cnr < 4; ++cnr) {
                    const int o = it.off[cnr];
                    if (o >= 0) {
                        ld8v(x + (long)o * a.C + c8 * 8, v[cnr]);
                        const float wq = it.w[cnr] * it.mask;
                        float* dst = dx + (long)o * a.C + c8 * 8;
#pragma unroll
                        for (int q = 0; q < 8; ++q) unsafeAtomicAdd(dst + q, wq * d[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[cnr][q] = 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float val = ((it.w[0] * v[0][q] + it.w[1] * v[1][q]) + it.w[2] * v[2][q]) + it.w[3] * v[3][q];
                    const float gy = (v[2][q] - v[0][q]) * it.hw + (v[3][q] - v[1][q]) * it.lw;
                    const float gx = (v[1][q] - v[0][q]) * it.hh + (v[3][q] - v[2][q]) * it.lh;
                    sdm += d[q] * val;
                    sdy += d[q] * gy;
                    sdx += d[q] * gx;
                }
                sdy *= it.mask; sdx *= it.mask;
            }
        }
        // reduce over the item's lane segment (consecutive lanes share an item: i = (p*K + k)*cg + c8)
        for (int o = seg >> 1; o > 0; o >>= 1) {
            sdy += __shfl_xor(sdy, o, 64);
            sdx += __shfl_xor(sdx, o, 64);
            sdm += __shfl_xor(sdm, o, 64);
        }
        if (live && ((threadIdx.x & (seg - 1)) == 0)) {
            if (cpg <= 64) {            // the segment is the whole item: plain store
                red[item_idx][0] = sdy; red[item_idx][1] = sdx; red[item_idx][2] = sdm;
            } else {                    // several segments per item: LDS atomics
                atomicAdd(&red[item_idx][0], sdy); atomicAdd(&red[item_idx][1], sdx); atomicAdd(&red[item_idx][2], sdm);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npix * KG; i += blockDim.x) {
        const int p = i / KG, rr = i - p * KG;
        const int g = rr / K, k = rr - g * K;
        const float mk = items[i].mask;
        float* o = dom + (m0 + p) * a.omc;
        o[g * 2 * K + 2 * k] = red[i][0];
        o[g * 2 * K + 2 * k + 1] = red[i][1];
        o[2 * KG + g * K + k] = red[i][2] * mk * (1.f - mk);
    }
}


// ---- backward, tiled: the fast path when every 64-channel chunk lies inside one deformable group ----------------------------
// Workgroup = (TH x TW tile of output pixels of one image) x (64-channel chunk).  Gradient contributions to x land in an fp32
// LDS window that covers the tile's receptive field plus a 2-pixel offset margin (lanes = channels, so every LDS atomic is a
// conflict-free 256-byte row); only contributions whose sample point left the window fall back to global atomics.  The window
// is flushed once with coalesced global atomics (neighbouring tiles overlap in their halos), which cuts the global atomic
// traffic of the generic kernel by ~10x and makes every global atomic a full 256-byte line pair.

// Wave64 sum on the DPP path (no LDS traffic): quad swaps, row mirrors, then row broadcasts; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(r);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_step<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_step<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_step<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_step<0x140, 0xf>(v);     // row_mirror: every lane of a row holds the row sum
    v = dpp_step<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
    v = dpp_step<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3
    return v;
}

constexpr int MDCN_WIN = 15;       // LDS window: at most 15 x 15 pixels x 64 channels x 4 B = 57.6 KB
constexpr int MDCN_MARGIN = 2;
constexpr int MDCN_TI = 8 * 8 * MDCN_MAXK;      // items of one tile (576)

struct TItem {
    int hw;                // (hl << 16) | (wl & 0xffff): top-left corner pixel; INT_MIN: sample outside the image
    float lh, lw, mask;
    int m;                 // flat output pixel index; -1: the pixel is outside the output map
};
constexpr int TITEM_OUTSIDE = (int)0x80000000;

// Tiled backward.  Workgroup = one TH x TW tile of output pixels of one image; it walks its share of the 64-channel chunks
// (blockIdx.y, stride gridDim.y).  Per chunk the gradient w.r.t. x is accumulated in an fp32 LDS window (tile receptive field +
// 2-pixel offset margin) whose columns are lane-private (lane = channel) and whose rows are wave-private (row & 7 == wave), so
// the accumulation is a plain LDS read-modify-write: no atomics (ds_add_f32 and, worse, global fp32 atomics at ~2.3 G
// transactions/s were the whole cost of the first versions).  The window is stored to the workspace and
// mdcn_window_gather_kernel sums the <= 2 x 2 overlapping windows per input pixel.  The offset / mask gradients are reduced
// over channels with DPP wave sums by the wave that owns the sample's top row and accumulated over the chunks in LDS; with
// gridDim.y > 1 the per-split partials go to the workspace and mdcn_dom_reduce_kernel adds them.
// Only samples that leave the window (|offset| beyond the margin) fall back to global atomics.
template <typename T>
__global__ __launch_bounds__(512) void mdcn_sample_bwd_tile_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                                   const T* __restrict__ dcol, float* __restrict__ dx,
                                                                   float* __restrict__ dom, const MdcnArgs a, int TH, int TW,
                                                                   int tiles_x, int tiles_y, float* __restrict__ ws,
                                                                   float* __restrict__ domp) {
    __shared__ float win[MDCN_WIN * MDCN_WIN * 64];
    __shared__ TItem items[MDCN_TI];
    __shared__ float red[MDCN_TI][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int K = a.kh * a.kw;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wy0 = oy0 * a.stride - a.pad - MDCN_MARGIN, wx0 = ox0 * a.stride - a.pad - MDCN_MARGIN;
    const int WH = (TH - 1) * a.stride + (a.kh - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int WW = (TW - 1) * a.stride + (a.kw - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int nitems = TH * TW * K;
    const int nchunks = a.C >> 6;
    const long img_base = (long)b * a.IH * a.IW;
    for (int i = tid; i < nitems; i += 512) {                   // items of the whole tile (deform_groups == 1 on this path)
        const int k = i % K;
        const int p = i / K;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        TItem it;
        it.m = -1; it.hw = TITEM_OUTSIDE; it.lh = it.lw = it.mask = 0.f;
        if (oy < a.OH && ox < a.OW) {
            const long m = ((long)b * a.OH + oy) * a.OW + ox;
            const float* rp = om + m * a.omc;
            const int ki = k / a.kw, kj = k - ki * a.kw;
            const float h = (float)(oy * a.stride - a.pad + ki * a.dil) + rp[2 * k];
            const float w = (float)(ox * a.stride - a.pad + kj * a.dil) + rp[2 * k + 1];
            const float hf = floorf(h), wf = floorf(w);
            it.m = (int)m;
            it.mask = 1.f / (1.f + expf(-rp[2 * K + k]));
            it.lh = h - hf; it.lw = w - wf;
            if ((h > -1.f) && (w > -1.f) && (h < (float)a.IH) && (w < (float)a.IW))
                it.hw = (int)(((unsigned)(int)hf << 16) | ((unsigned)(int)wf & 0xffffu));
        }
        items[i] = it;
        red[i][0] = 0.f; red[i][1] = 0.f; red[i][2] = 0.f;
    }
    for (int chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        const int c = chunk * 64 + lane;
        __syncthreads();                                        // items ready / previous window stored
        for (int i = tid; i < WH * WW * 64; i += 512) win[i] = 0.f;
        __syncthreads();
        for (int ii = 0; ii < nitems; ++ii) {                   // every wave scans all items (wave-uniform control flow)
            const TItem it = items[ii];
            if (it.hw == TITEM_OUTSIDE) continue;
            const int hl = it.hw >> 16, wl = (int)(short)(it.hw & 0xffff);
            const bool top = ((hl - wy0) & 7) == wave, bot = ((hl + 1 - wy0) & 7) == wave;
            if (!(top || bot)) continue;
            const int k = ii % K;
            const float d = Elem<T>::ld(dcol + ((long)it.m * K + k) * a.C + c);
            const float lh = it.lh, lw = it.lw, hh = 1.f - lh, hw = 1.f - lw, mk = it.mask;
            const float wgt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (top) {
#pragma unroll
                for (int cnr = 0; cnr < 4; ++cnr) {
                    const int py = hl + (cnr >> 1), px = wl + (cnr & 1);
                    if ((py >= 0) && (py <= a.IH - 1) && (px >= 0) && (px <= a.IW - 1))
                        v[cnr] = Elem<T>::ld(x + (img_base + (long)py * a.IW + px) * a.C + c);
                }
            }
#pragma unroll
            for (int cnr = 0; cnr < 4; ++cnr) {
                if (!((cnr >> 1) ? bot : top)) continue;
                const int py = hl + (cnr >> 1), px = wl + (cnr & 1);
                if ((py >= 0) && (py <= a.IH - 1) && (px >= 0) && (px <= a.IW - 1) && wgt[cnr] != 0.f) {
                    const float contrib = wgt[cnr] * mk * d;
                    const int wy = py - wy0, wx = px - wx0;
                    if (wy >= 0 && wy < WH && wx >= 0 && wx < WW) win[(wy * WW + wx) * 64 + lane] += contrib;
                    else unsafeAtomicAdd(dx + (img_base + (long)py * a.IW + px) * a.C + c, contrib);
                }
            }
            if (top) {
                const float sdm = wave_sum(d * (((wgt[0] * v[0] + wgt[1] * v[1]) + wgt[2] * v[2]) + wgt[3] * v[3]));
                const float sdy = wave_sum(d * mk * ((v[2] - v[0]) * hw + (v[3] - v[1]) * lw));
                const float sdx = wave_sum(d * mk * ((v[1] - v[0]) * hh + (v[3] - v[2]) * lh));
                if (lane == 63) { red[ii][0] += sdy; red[ii][1] += sdx; red[ii][2] += sdm * mk * (1.f - mk); }
            }
        }
        __syncthreads();
        float* wp = ws + (long)blockIdx.x * (WH * WW) * a.C + chunk * 64;              // workspace layout [tile][pixel][C]
        for (int i = tid * 4; i < WH * WW * 64; i += 512 * 4) st4(wp + (long)(i >> 6) * a.C + (i & 63), win + i);
    }
    __syncthreads();
    float* dst = gridDim.y == 1 ? dom : domp + (long)blockIdx.y * a.M * a.omc;
    for (int i = tid; i < nitems * 3; i += 512) {
        const int ii = i / 3, q = i - ii * 3;
        const int m = items[ii].m;
        if (m >= 0) {
            const int k = ii % K;
            const int ch = q == 0 ? 2 * k : (q == 1 ? 2 * k + 1 : 2 * K + k);
            dst[(long)m * a.omc + ch] = red[ii][q];
        }
    }
}


// ---- backward, binned (bf16; the training path) -------------------------------------------------------------------------------
// The window kernel above spends its time on per-item bookkeeping replicated over 64 lanes and repeated per 64-channel chunk
// (measured: ~100 GB/s).  Here the bookkeeping is done ONCE per tile and the channel work is pure streaming:
//   1. items: (pixel, tap) -> corner pixel, bilinear fractions, mask                       (as above)
//   2. binning: every (item, corner) contribution is appended to the list of the window pixel it lands on
//      (counting sort in LDS: integer ds_add, a 225-entry scan, cursor fill)               -- integer LDS atomics only
//   3. dx: each wave walks the window pixels it owns; per list entry it reads the item's dcol row (lane = CPL consecutive
//      channels, one coalesced 64*CPL*2-byte row) and accumulates weight * row in registers; one store of the finished
//      pixel to the workspace window.  No atomics, no LDS traffic for the channels.
//   4. offset / mask gradients: each wave walks items; dcol row + the four corner rows of x, per-lane partial dot products,
//      three DPP wave sums, one store.  Corners that left the window are scattered here with global atomics (rare).
// CPL (channels per lane) in {1,2,4,8}: one workgroup covers a 64*CPL-channel slab; C/(64*CPL) slabs ride on blockIdx.y.
template <int CPL> struct RowLd;
template <> struct RowLd<1> { static __device__ __forceinline__ void ld(const bf16_t* p, float* v) { v[0] = bf16_to_f32(*p); } };
template <> struct RowLd<2> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        unpack2_16(t, v[0], v[1]);
    }
};
template <> struct RowLd<4> { static __device__ __forceinline__ void ld(const bf16_t* p, float* v) { ld4(p, v); } };
template <> struct RowLd<8> { static __device__ __forceinline__ void ld(const bf16_t* p, float* v) { ld8v(p, v); } };

struct BinEntry { int row; float wt; };     // row = m*K + k (index of the dcol row), wt = bilinear weight * mask

template <int CPL>
__global__ __launch_bounds__(256) void mdcn_sample_bwd_bin_kernel(const bf16_t* __restrict__ x, const float* __restrict__ om,
                                                                  const bf16_t* __restrict__ dcol, float* __restrict__ dx,
                                                                  float* __restrict__ dom, const MdcnArgs a, int TH, int TW,
                                                                  int tiles_x, int tiles_y, float* __restrict__ ws,
                                                                  float* __restrict__ domp) {
    constexpr int NW = 4;
    constexpr int NPIX = MDCN_WIN * MDCN_WIN;
    __shared__ TItem items[MDCN_TI];
    __shared__ BinEntry entries[MDCN_TI * 4];
    __shared__ int cnt[NPIX + 1], start[NPIX + 1];
    __shared__ int far_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int K = a.kh * a.kw;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wy0 = oy0 * a.stride - a.pad - MDCN_MARGIN, wx0 = ox0 * a.stride - a.pad - MDCN_MARGIN;
    const int WH = (TH - 1) * a.stride + (a.kh - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int WW = (TW - 1) * a.stride + (a.kw - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int npix = WH * WW;
    const int nitems = TH * TW * K;
    const long img_base = (long)b * a.IH * a.IW;
    const int cb = blockIdx.y * 64 * CPL + lane * CPL;          // first channel of this lane
    // ---- 1. items
    for (int i = tid; i < nitems; i += 256) {
        const int k = i % K;
        const int p = i / K;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        TItem it;
        it.m = -1; it.hw = TITEM_OUTSIDE; it.lh = it.lw = it.mask = 0.f;
        if (oy < a.OH && ox < a.OW) {
            const long m = ((long)b * a.OH + oy) * a.OW + ox;
            const float* rp = om + m * a.omc;
            const int ki = k / a.kw, kj = k - ki * a.kw;
            const float h = (float)(oy * a.stride - a.pad + ki * a.dil) + rp[2 * k];
            const float w = (float)(ox * a.stride - a.pad + kj * a.dil) + rp[2 * k + 1];
            const float hf = floorf(h), wf = floorf(w);
            it.m = (int)m;
            it.mask = 1.f / (1.f + expf(-rp[2 * K + k]));
            it.lh = h - hf; it.lw = w - wf;
            if ((h > -1.f) && (w > -1.f) && (h < (float)a.IH) && (w < (float)a.IW))
                it.hw = (int)(((unsigned)(int)hf << 16) | ((unsigned)(int)wf & 0xffffu));
        }
        items[i] = it;
    }
    for (int i = tid; i <= npix; i += 256) cnt[i] = 0;
    if (tid == 0) far_flag = 0;
    __syncthreads();
    // ---- 2. binning (thread per item x corner); pass 0 counts, pass 1 fills
    auto corner = [&](int i, int& pix, float& wt) -> int {      // 0: nothing, 1: in window, 2: valid but outside the window
        const TItem it = items[i >> 2];
        if (it.hw == TITEM_OUTSIDE) return 0;
        const int cnr = i & 3;
        const int py = (it.hw >> 16) + (cnr >> 1), px = (int)(short)(it.hw & 0xffff) + (cnr & 1);
        if (py < 0 || py > a.IH - 1 || px < 0 || px > a.IW - 1) return 0;
        const float fy = (cnr >> 1) ? it.lh : 1.f - it.lh, fx = (cnr & 1) ? it.lw : 1.f - it.lw;
        wt = fy * fx;
        if (wt == 0.f) return 0;
        wt *= it.mask;
        const int wy = py - wy0, wx = px - wx0;
        if (wy < 0 || wy >= WH || wx < 0 || wx >= WW) return 2;
        pix = wy * WW + wx;
        return 1;
    };
    for (int i = tid; i < nitems * 4; i += 256) {
        int pix = 0; float wt = 0.f;
        const int r = corner(i, pix, wt);
        if (r == 1) atomicAdd(&cnt[pix], 1);
        else if (r == 2) far_flag = 1;
    }
    __syncthreads();
    if (tid < 64) {                                              // exclusive scan of <= 225 counters by one wave
        int run = 0;
        for (int base = 0; base < npix; base += 64) {
            const int i = base + lane;
            int v = i < npix ? cnt[i] : 0, incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            if (i < npix) { start[i] = run + incl - v; cnt[i] = run + incl - v; }      // cnt becomes the fill cursor
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) start[npix] = run;
    }
    __syncthreads();
    for (int i = tid; i < nitems * 4; i += 256) {
        int pix = 0; float wt = 0.f;
        if (corner(i, pix, wt) == 1) {
            const int pos = atomicAdd(&cnt[pix], 1);
            BinEntry e;
            e.row = items[i >> 2].m * K + (i >> 2) % K;
            e.wt = wt;
            entries[pos] = e;
        }
    }
    __syncthreads();
    // ---- 3. dx: window pixels, lists in registers-accumulate form
    float* wsb = ws + (long)blockIdx.x * npix * a.C;
    for (int p = wave; p < npix; p += NW) {
        const int s0 = __builtin_amdgcn_readfirstlane(start[p]), s1 = __builtin_amdgcn_readfirstlane(start[p + 1]);
        float acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
        for (int e = s0; e < s1; e += 4) {
            float dv[4][CPL], wq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int eu = (e + u < s1) ? e + u : s1 - 1;
                const BinEntry en = entries[eu];
                const int row = __builtin_amdgcn_readfirstlane(en.row);
                wq[u] = (e + u < s1) ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(en.wt))) : 0.f;
                RowLd<CPL>::ld(dcol + (long)row * a.C + cb, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < CPL; ++q) acc[q] += wq[u] * dv[u][q];
        }
        float* o = wsb + (long)p * a.C + cb;
        if (CPL == 1) o[0] = acc[0];
        else if (CPL == 2) *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[1]);
        else {
#pragma unroll
            for (int q = 0; q < CPL; q += 4) st4(o + q, acc + q);
        }
    }
    // ---- 4. offset / mask gradients (and the rare out-of-window scatter)
    const bool any_far = far_flag != 0;
    float* dstp = gridDim.y == 1 ? dom : domp + (long)blockIdx.y * a.M * a.omc;
    for (int ii = wave; ii < nitems; ii += NW) {
        const TItem it = items[ii];
        const int m = __builtin_amdgcn_readfirstlane(it.m);
        if (m < 0) continue;
        const int k = ii % K;
        const int hwp = __builtin_amdgcn_readfirstlane(it.hw);
        float sdy = 0.f, sdx = 0.f, sdm = 0.f;
        const float mk = it.mask;
        if (hwp != TITEM_OUTSIDE) {
            const int hl = hwp >> 16, wl = (int)(short)(hwp & 0xffff);
            const float lh = it.lh, lw = it.lw, hh = 1.f - lh, hw = 1.f - lw;
            const float wgt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            float d[CPL], v[4][CPL];
            RowLd<CPL>::ld(dcol + ((long)m * K + k) * a.C + cb, d);
#pragma unroll
            for (int cnr = 0; cnr < 4; ++cnr) {
                const int py = hl + (cnr >> 1), px = wl + (cnr & 1);
                if ((py >= 0) && (py <= a.IH - 1) && (px >= 0) && (px <= a.IW - 1)) {
                    RowLd<CPL>::ld(x + (img_base + (long)py * a.IW + px) * a.C + cb, v[cnr]);
                    if (any_far && wgt[cnr] != 0.f) {
                        const int wy = py - wy0, wx = px - wx0;
                        if (wy < 0 || wy >= WH || wx < 0 || wx >= WW) {
                            float* g = dx + (img_base + (long)py * a.IW + px) * a.C + cb;
#pragma unroll
                            for (int q = 0; q < CPL; ++q) unsafeAtomicAdd(g + q, wgt[cnr] * mk * d[q]);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) v[cnr][q] = 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                sdm += d[q] * (((wgt[0] * v[0][q] + wgt[1] * v[1][q]) + wgt[2] * v[2][q]) + wgt[3] * v[3][q]);
                sdy += d[q] * ((v[2][q] - v[0][q]) * hw + (v[3][q] - v[1][q]) * lw);
                sdx += d[q] * ((v[1][q] - v[0][q]) * hh + (v[3][q] - v[2][q]) * lh);
            }
            sdm = wave_sum(sdm); sdy = wave_sum(sdy * mk); sdx = wave_sum(sdx * mk);
        }
        if (lane == 63) {
            float* o = dstp + (long)m * a.omc;
            o[2 * k] = sdy; o[2 * k + 1] = sdx; o[2 * K + k] = sdm * mk * (1.f - mk);
        }
    }
}

__global__ void mdcn_dom_reduce_kernel(const float* __restrict__ domp, float* __restrict__ dom, long n, int splits, int omc,
                                       int nch) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        if ((int)(i % omc) < nch)                               // padding channels of the offset tensor hold no partials
            for (int s2 = 0; s2 < splits; ++s2) acc += domp[s2 * n + i];
        dom[i] = acc;
    }
}

// Second pass of the tiled backward: dx[b,y,x,c] += sum over the (at most 2 x 2) tile windows that contain input pixel (y,x).
__global__ __launch_bounds__(256) void mdcn_window_gather_kernel(const float* __restrict__ ws, float* __restrict__ dx,
                                                                 const MdcnArgs a, int TH, int TW, int tiles_x, int tiles_y) {
    const int WH = (TH - 1) * a.stride + (a.kh - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int WW = (TW - 1) * a.stride + (a.kw - 1) * a.dil + 2 * MDCN_MARGIN + 2;
    const int sy = TH * a.stride, sx = TW * a.stride, org = a.pad + MDCN_MARGIN;
    const int cv = a.C >> 2;
    const long total = (long)a.B * a.IH * a.IW * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cv);
        long p = i / cv;
        const int xx = (int)(p % a.IW); p /= a.IW;
        const int yy = (int)(p % a.IH);
        const int b = (int)(p / a.IH);
        // tiles ty with  ty*sy - org <= yy < ty*sy - org + WH
        int ty1 = (yy + org) / sy, ty0 = (yy + org - WH + sy) / sy;      // floor / ceil((yy+org-WH+1)/sy) for non-negatives
        if (yy + org - WH + 1 <= 0) ty0 = 0;
        int tx1 = (xx + org) / sx, tx0 = (xx + org - WW + sx) / sx;
        if (xx + org - WW + 1 <= 0) tx0 = 0;
        if (ty1 > tiles_y - 1) ty1 = tiles_y - 1;
        if (tx1 > tiles_x - 1) tx1 = tiles_x - 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
                const int wy = yy - (ty * sy - org), wx = xx - (tx * sx - org);
                const long blk = ((long)b * tiles_y + ty) * tiles_x + tx;
                float v[4];
                ld4(ws + (blk * (WH * WW) + wy * WW + wx) * (long)a.C + c4 * 4, v);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += v[q];
            }
        float o[4];
        ld4(dx + i * 4, o);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] += acc[q];
        st4(dx + i * 4, o);
    }
}

int mdcn_check(const MdcnArgs& a) {
    const int K = a.kh * a.kw;
    if (K < 1 || K > MDCN_MAXK || a.DG < 1 || a.DG > MDCN_MAXDG || (a.C % (8 * a.DG)) || a.omc < 3 * K * a.DG)
        return (int)hipErrorInvalidValue;
    const int cpg = (a.C >> 3) / a.DG;
    if (cpg & (cpg - 1)) return (int)hipErrorInvalidValue;      // lane-segment reduction wants a power of two
    if ((long)a.B * a.IH * a.IW > 0x7fffffffL) return (int)hipErrorInvalidValue;
    return 0;
}

}  // namespace

LOFT_EXPORT int loft_mdcn_sample_fwd(const void* x, const float* offmask, void* col, int dtype, int B, int IH, int IW, int C,
                                     int OH, int OW, int kh, int kw, int stride, int pad, int dil, int deform_groups,
                                     int offmask_stride, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    MdcnArgs a{B, IH, IW, C, OH, OW, kh, kw, stride, pad, dil, deform_groups, offmask_stride, (long)B * OH * OW};
    if (int e = mdcn_check(a)) return e;
    if (a.M <= 0) return 0;
    dim3 grid((unsigned)((a.M + MDCN_PIX - 1) / MDCN_PIX));
    if (dtype == LOFT_F32)
        hipLaunchKernelGGL(mdcn_sample_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, offmask,
                           (float*)col, a);
    else
        hipLaunchKernelGGL(mdcn_sample_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, offmask,
                           (bf16_t*)col, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

struct MdcnTiling { int TH, TW, tiles_x, tiles_y, WH, WW, splits, cpl, nslab; long win_floats, domp_floats; };

static bool mdcn_tiling(int B, int C, int OH, int OW, int kh, int kw, int stride, int dil, int deform_groups, int omc,
                        MdcnTiling* t) {
    if (deform_groups != 1 || C % 64 != 0 || dil != 1) return false;
    // 8x8 output tiles at stride 1, 4x4 at stride 2: the receptive field + margin fits the 15x15 LDS window
    t->TH = t->TW = (stride == 1 ? 8 : 4);
    t->WH = (t->TH - 1) * stride + (kh - 1) * dil + 2 * MDCN_MARGIN + 2;
    t->WW = (t->TW - 1) * stride + (kw - 1) * dil + 2 * MDCN_MARGIN + 2;
    if (t->WH > MDCN_WIN || t->WW > MDCN_WIN) return false;
    t->tiles_x = (OW + t->TW - 1) / t->TW;
    t->tiles_y = (OH + t->TH - 1) / t->TH;
    const long tiles = (long)B * t->tiles_x * t->tiles_y;
    const int nchunks = C / 64;
    long want = (1024 + tiles - 1) / tiles;                     // >= ~1024 workgroups; all chunks in one when tiles suffice
    t->splits = (int)(want < 1 ? 1 : (want > nchunks ? nchunks : want));
    // binned bf16 kernel: channels per lane (a workgroup covers a 64*cpl-channel slab); smaller slabs when tiles are few
    t->cpl = nchunks >= 8 ? 8 : (nchunks >= 4 ? 4 : (nchunks >= 2 ? 2 : 1));
    while (nchunks % t->cpl) t->cpl >>= 1;
    while (t->cpl > 1 && tiles * (nchunks / t->cpl) < 1024) t->cpl >>= 1;
    t->nslab = nchunks / t->cpl;
    t->win_floats = tiles * nchunks * (long)(t->WH * t->WW * 64);
    const int parts = t->splits > t->nslab ? t->splits : t->nslab;
    t->domp_floats = parts > 1 ? (long)parts * B * OH * OW * omc : 0;
    return true;
}

LOFT_EXPORT int64_t loft_mdcn_bwd_workspace_bytes(int B, int C, int OH, int OW, int kh, int kw, int stride, int dil,
                                                  int deform_groups, int offmask_stride) {
    MdcnTiling t;
    if (!mdcn_tiling(B, C, OH, OW, kh, kw, stride, dil, deform_groups, offmask_stride, &t)) return 0;
    return (int64_t)(t.win_floats + t.domp_floats) * (int64_t)sizeof(float);
}

LOFT_EXPORT int loft_mdcn_sample_bwd(const void* x, const float* offmask, const void* dcol, float* dx, float* doffmask,
                                     int dtype, int B, int IH, int IW, int C, int OH, int OW, int kh, int kw, int stride,
                                     int pad, int dil, int deform_groups, int offmask_stride, void* workspace,
                                     void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    MdcnArgs a{B, IH, IW, C, OH, OW, kh, kw, stride, pad, dil, deform_groups, offmask_stride, (long)B * OH * OW};
    if (int e = mdcn_check(a)) return e;
    if (a.M <= 0) return 0;
    MdcnTiling t;
    if (workspace && mdcn_tiling(B, C, OH, OW, kh, kw, stride, dil, deform_groups, offmask_stride, &t)) {
        const unsigned ntiles = (unsigned)(t.tiles_x * t.tiles_y * B);
        float* ws = (float*)workspace;
        float* domp = ws + t.win_floats;
        int parts = t.splits;
        if (dtype == LOFT_F32)
            hipLaunchKernelGGL(mdcn_sample_bwd_tile_kernel<float>, dim3(ntiles, t.splits), dim3(512), 0, (hipStream_t)stream,
                               (const float*)x, offmask, (const float*)dcol, dx, doffmask, a, t.TH, t.TW, t.tiles_x, t.tiles_y, ws,
                               domp);
        else {
            parts = t.nslab;
            const dim3 bg(ntiles, t.nslab);
#define LOFT_BIN_LAUNCH(CPL)                                                                                                  \
    hipLaunchKernelGGL(mdcn_sample_bwd_bin_kernel<CPL>, bg, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, offmask,    \
                       (const bf16_t*)dcol, dx, doffmask, a, t.TH, t.TW, t.tiles_x, t.tiles_y, ws, domp)
            if (t.cpl == 8) LOFT_BIN_LAUNCH(8);
            else if (t.cpl == 4) LOFT_BIN_LAUNCH(4);
            else if (t.cpl == 2) LOFT_BIN_LAUNCH(2);
            else LOFT_BIN_LAUNCH(1);
#undef LOFT_BIN_LAUNCH
        }
        LOFT_LAUNCH_CHECK();
        const long nvec = (long)B * IH * IW * (C / 4);
        long blocks = (nvec + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(mdcn_window_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, dx, a, t.TH,
                           t.TW, t.tiles_x, t.tiles_y);
        LOFT_LAUNCH_CHECK();
        if (parts > 1) {
            const long n = a.M * a.omc;
            long rb = (n + 255) / 256;
            if (rb > 8192) rb = 8192;
            hipLaunchKernelGGL(mdcn_dom_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, domp, doffmask, n,
                               parts, a.omc, 3 * kh * kw);
            LOFT_LAUNCH_CHECK();
        }
        return 0;
    }
    dim3 grid((unsigned)((a.M + MDCN_PIX - 1) / MDCN_PIX));
    if (dtype == LOFT_F32)
        hipLaunchKernelGGL(mdcn_sample_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, offmask,
                           (const float*)dcol, dx, doffmask, a);
    else
        hipLaunchKernelGGL(mdcn_sample_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, offmask,
                           (const bf16_t*)dcol, dx, doffmask, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}
