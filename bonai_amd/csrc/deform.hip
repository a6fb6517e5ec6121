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
// Backward: given dcol, produce d(raw) (offsets and mask logits, including the sigmoid derivative) and scatter
// d(x) += mask * w_corner * dcol into an fp32 gradient map with hardware fp32 atomics; the per-item channel reductions run
// as cross-lane shuffles inside the item's lane segment.
#include "loft_common.h"
#include "../../include/loft_hip.h"

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
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
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

LOFT_EXPORT int loft_mdcn_sample_bwd(const void* x, const float* offmask, const void* dcol, float* dx, float* doffmask,
                                     int dtype, int B, int IH, int IW, int C, int OH, int OW, int kh, int kw, int stride,
                                     int pad, int dil, int deform_groups, int offmask_stride, void* stream) {
    MdcnArgs a{B, IH, IW, C, OH, OW, kh, kw, stride, pad, dil, deform_groups, offmask_stride, (long)B * OH * OW};
    if (int e = mdcn_check(a)) return e;
    if (a.M <= 0) return 0;
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
