// resample.hip -- HBM-bound resampling / fusion kernels of the HRNet-W32 + HRFPN path (BASELINE config 5), gfx950, NHWC,
// 8-channel (16-byte bf16 / 32-byte fp32) vector accesses.
//
//   fuse_sum_relu     : HRModule.forward fuse step  y_i = relu(sum_j f_ij(x_j))  (mmdet/models/backbones/hrnet.py:177-195):
//                       the terms arrive at their own resolution; coarser ones are nearest-upsampled by 2^shift on the fly
//                       (nn.Upsample(mode='nearest') of hrnet.py:141-143) -- one pass instead of (#terms - 1) adds + upsamples + relu.
//   blocksum_masked   : its backward per coarse term: g_j = sum over the 2^shift x 2^shift block of g * (y > 0).
//   bilinear_up_slot  : HRFPN.forward  F.interpolate(x_i, scale_factor=2^i, mode='bilinear')  (align_corners=False) written
//                       straight into its channel slot of the concatenated tensor (mmdet/models/necks/hrfpn.py:79-85), + backward.
//   avgpool           : F.avg_pool2d(out, 2^i, 2^i) pyramid (hrfpn.py:90-92) + backward.
//   stem3x3s2         : HRNet stem conv1 3x3/2 (3 -> 64) + frozen-stat BN + ReLU from the fp32 NCHW image (hrnet.py:273-281,
//                       481-483) and the weight-gradient sums of that conv (the image needs no gradient).
// Roofline for all of these: HBM bandwidth (bytes = each operand once).
#include "loft_common.h"
#include "../../include/loft_hip.h"

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float v[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack8_16(t, v);
}
__device__ __forceinline__ void ld8(const float* p, float v[8]) { ld4(p, v); ld4(p + 4, v + 4); }
__device__ __forceinline__ void st8(bf16_t* p, const float v[8]) { st4(p, v); st4(p + 4, v + 4); }
__device__ __forceinline__ void st8(float* p, const float v[8]) { st4(p, v); st4(p + 4, v + 4); }

inline dim3 grid_for(long nvec) {
    long b = (nvec + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

struct FuseArgs {
    const void* src[4];
    int shift[4];
    int n, B, H, W, C, relu;
};

template <typename T>
__global__ void fuse_sum_relu_kernel(const FuseArgs a, T* __restrict__ out) {
    const int cg = a.C >> 3;
    const long nvec = (long)a.B * a.H * a.W * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H);
        const int b = (int)(p / a.H);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < a.n) {
                const int sh = a.shift[j];
                const long e = ((((long)b * (a.H >> sh)) + (y >> sh)) * (a.W >> sh) + (x >> sh)) * a.C + c8 * 8;
                float v[8];
                ld8(reinterpret_cast<const T*>(a.src[j]) + e, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += v[q];
            }
        }
        if (a.relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fmaxf(acc[q], 0.f);
        }
        st8(out + i * 8, acc);
    }
}

// out[b,i,j,:] = sum over the (1<<sh)^2 block of g[b, i<<sh + dy, j<<sh + dx, :] * (y > 0)     (y == NULL: no mask)
template <typename T>
__global__ void blocksum_masked_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ out, int B, int Hc,
                                       int Wc, int C, int sh) {
    const int cg = C >> 3, f = 1 << sh;
    const int H = Hc << sh, W = Wc << sh;
    const long nvec = (long)B * Hc * Wc * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int xj = (int)(p % Wc); p /= Wc;
        const int yi = (int)(p % Hc);
        const int b = (int)(p / Hc);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                const long e = (((long)b * H + (yi << sh) + dy) * W + (xj << sh) + dx) * C + c8 * 8;
                float gv[8];
                ld8(g + e, gv);
                if (y) {
                    float yv[8];
                    ld8(y + e, yv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) gv[q] = yv[q] > 0.f ? gv[q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += gv[q];
            }
        st8(out + i * 8, acc);
    }
}

// PyTorch upsample_bilinear2d, align_corners=False, integer scale s = 1 << sh:
//   src = max((dst + 0.5) / s - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, n - 1); l1 = src - i0; l0 = 1 - l1
__device__ __forceinline__ void bil_coord(int d, int sh, int n, int& i0, int& i1, float& l0, float& l1) {
    const float s = fmaxf(((float)d + 0.5f) * (1.0f / (float)(1 << sh)) - 0.5f, 0.f);
    i0 = (int)s;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

template <typename T>
__global__ void bilinear_up_slot_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int h, int w, int C, int sh,
                                        int Ctot, int coff) {
    const int cg = C >> 3, H = h << sh, W = w << sh;
    const long nvec = (long)B * H * W * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bil_coord(y, sh, h, y0, y1, ly0, ly1);
        bil_coord(x, sh, w, x0, x1, lx0, lx1);
        float v00[8], v01[8], v10[8], v11[8], o[8];
        const T* sb = src + (long)b * h * w * C + c8 * 8;
        ld8(sb + ((long)y0 * w + x0) * C, v00); ld8(sb + ((long)y0 * w + x1) * C, v01);
        ld8(sb + ((long)y1 * w + x0) * C, v10); ld8(sb + ((long)y1 * w + x1) * C, v11);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = ly0 * (lx0 * v00[q] + lx1 * v01[q]) + ly1 * (lx0 * v10[q] + lx1 * v11[q]);
        st8(dst + (((long)b * H + y) * W + x) * Ctot + coff + c8 * 8, o);
    }
}

// gradient w.r.t. src: gather over the destination pixels that read source pixel (i, j)
template <typename T>
__global__ void bilinear_up_slot_bwd_kernel(const T* __restrict__ g, T* __restrict__ dsrc, int B, int h, int w, int C, int sh,
                                            int Ctot, int coff) {
    const int cg = C >> 3, H = h << sh, W = w << sh, s = 1 << sh;
    const long nvec = (long)B * h * w * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int sj = (int)(p % w); p /= w;
        const int si = (int)(p % h);
        const int b = (int)(p / h);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        const int ylo = max(0, (si - 1) * s), yhi = min(H - 1, (si + 2) * s - 1);
        const int xlo = max(0, (sj - 1) * s), xhi = min(W - 1, (sj + 2) * s - 1);
        for (int y = ylo; y <= yhi; ++y) {
            int y0, y1; float ly0, ly1;
            bil_coord(y, sh, h, y0, y1, ly0, ly1);
            const float wy = (y0 == si ? ly0 : 0.f) + (y1 == si ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                int x0, x1; float lx0, lx1;
                bil_coord(x, sh, w, x0, x1, lx0, lx1);
                const float wx = (x0 == sj ? lx0 : 0.f) + (x1 == sj ? lx1 : 0.f);
                if (wx == 0.f) continue;
                float gv[8];
                ld8(g + (((long)b * H + y) * W + x) * Ctot + coff + c8 * 8, gv);
                const float wgt = wy * wx;
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += wgt * gv[q];
            }
        }
        st8(dsrc + i * 8, acc);
    }
}

template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Ho, int Wo, int C, int sh) {
    const int cg = C >> 3, f = 1 << sh, H = Ho << sh, W = Wo << sh;
    const float inv = 1.f / (float)(f * f);
    const long nvec = (long)B * Ho * Wo * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int xj = (int)(p % Wo); p /= Wo;
        const int yi = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                float v[8];
                ld8(src + (((long)b * H + (yi << sh) + dy) * W + (xj << sh) + dx) * C + c8 * 8, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += v[q];
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] *= inv;
        st8(dst + i * 8, acc);
    }
}

// dsrc (fine) += g (coarse) / f^2 broadcast over the f x f block   (accumulate != 0: add to the existing contents)
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ g, T* __restrict__ dsrc, int B, int Ho, int Wo, int C, int sh,
                                   int accumulate) {
    const int cg = C >> 3, f = 1 << sh, H = Ho << sh, W = Wo << sh;
    const float inv = 1.f / (float)(f * f);
    const long nvec = (long)B * H * W * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float gv[8], o[8];
        ld8(g + (((long)b * Ho + (y >> sh)) * Wo + (x >> sh)) * C + c8 * 8, gv);
        if (accumulate) ld8(dsrc + i * 8, o);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (accumulate ? o[q] : 0.f) + gv[q] * inv;
        st8(dsrc + i * 8, o);
    }
}

// ---- HRNet stem conv1: 3x3 stride 2 pad 1, 3 -> 64, + folded BN + ReLU.  Block = 16 x 16 output pixels.
template <typename T>
__global__ __launch_bounds__(256) void stem3x3s2_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
    __shared__ float patch[3][33][34];
    __shared__ float wl[27][64];   // [k][oc], k = (c, r, s)
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * 64; i += 256) {
        const int oc = i / 27, k = i - oc * 27;
        wl[k][oc] = w[i];
    }
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
    for (int i = tid; i < 3 * 33 * 33; i += 256) {
        const int c = i / (33 * 33), r = i - c * 33 * 33;
        const int py = r / 33, px = r - py * 33;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((long)b * 3 + c) * H + iy) * W + ix];
        patch[c][py][px] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= Ho || ox >= Wo) return;
    T* op = out + (((long)b * Ho + oy) * Wo + ox) * 64;
    for (int pass = 0; pass < 4; ++pass) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float v = patch[c][ty * 2 + r][tx * 2 + s];
                    const float* wk = &wl[(c * 3 + r) * 3 + s][pass * 16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = fmaf(v, wk[q], acc[q]);
                }
        float o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int oc = pass * 16 + q;
            o[q] = fmaxf(acc[q] * scale[oc] + shift[oc], 0.f);
        }
        st8(op + pass * 16, o);
        st8(op + pass * 16 + 8, o + 8);
    }
}

// S[t][n][c] += sum_pixels dpre[m][n] * patch[m][(c, t)],  db[n] += sum_pixels dpre[m][n],  dpre = g * (y > 0)
// (the gradient of the BN-folded weight in the [tap][Cout][Cin] packing loft_fold_unpack_bwd consumes).
// Thread (n = tid & 63, kq = tid >> 6) owns output channel n and the taps k = kq, kq + 4, ... of the 27.
template <typename T>
__global__ __launch_bounds__(256) void stem3x3s2_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ g,
                                                              const T* __restrict__ y, float* __restrict__ S,
                                                              float* __restrict__ db, int B, int H, int W, int Ho, int Wo,
                                                              int tiles_x, int tiles_y) {
    __shared__ float patch[3][33][34];
    __shared__ float dpre[256][65];      // +1: conflict-free column reads
    const int tid = threadIdx.x, n = tid & 63, kq = tid >> 6;
    float acc[7], accb = 0.f;
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = 0.f;
    const int ntiles = tiles_x * tiles_y * B;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tt = t;
        const int tx0 = tt % tiles_x; tt /= tiles_x;
        const int ty0 = tt % tiles_y;
        const int b = tt / tiles_y;
        const int oy0 = ty0 * 16, ox0 = tx0 * 16;
        const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
        __syncthreads();
        for (int i = tid; i < 3 * 33 * 33; i += 256) {
            const int c = i / (33 * 33), r = i - c * 33 * 33;
            const int py = r / 33, px = r - py * 33;
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((long)b * 3 + c) * H + iy) * W + ix];
            patch[c][py][px] = v;
        }
        for (int i = tid; i < 256 * 8; i += 256) {     // 256 pixels x 8 groups of 8 channels
            const int px = i >> 3, c8 = i & 7;
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            float gv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) gv[q] = 0.f;
            if (oy < Ho && ox < Wo) {
                const long e = (((long)b * Ho + oy) * Wo + ox) * 64 + c8 * 8;
                float yv[8];
                ld8(g + e, gv); ld8(y + e, yv);
#pragma unroll
                for (int q = 0; q < 8; ++q) gv[q] = yv[q] > 0.f ? gv[q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) dpre[px][c8 * 8 + q] = gv[q];
        }
        __syncthreads();
        for (int px = 0; px < 256; ++px) {
            const float d = dpre[px][n];
            const int py2 = (px >> 4) * 2, px2 = (px & 15) * 2;
            accb += d;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int k = kq + 4 * q;
                if (k < 27) {
                    const int c = k / 9, rs = k - c * 9, r = rs / 3, s = rs - r * 3;
                    acc[q] = fmaf(d, patch[c][py2 + r][px2 + s], acc[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int k = kq + 4 * q;
        if (k < 27) {
            const int c = k / 9, t = k - c * 9;
            unsafeAtomicAdd(S + ((long)t * 64 + n) * 3 + c, acc[q]);
        }
    }
    if (kq == 0) unsafeAtomicAdd(db + n, accb);
}

}  // namespace

LOFT_EXPORT int loft_fuse_sum_relu(const void* const* terms, const int* shifts, int n_terms, void* out, int dtype, int B, int H,
                                   int W, int C, int relu, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if (n_terms < 1 || n_terms > 4 || (C % 8)) return (int)hipErrorInvalidValue;
    FuseArgs a;
    for (int j = 0; j < 4; ++j) { a.src[j] = j < n_terms ? terms[j] : nullptr; a.shift[j] = j < n_terms ? shifts[j] : 0; }
    for (int j = 0; j < n_terms; ++j)
        if (shifts[j] < 0 || (H & ((1 << shifts[j]) - 1)) || (W & ((1 << shifts[j]) - 1))) return (int)hipErrorInvalidValue;
    a.n = n_terms; a.B = B; a.H = H; a.W = W; a.C = C; a.relu = relu;
    const long nvec = (long)B * H * W * (C / 8);
    if (nvec <= 0) return 0;
    if (dtype == LOFT_F32) hipLaunchKernelGGL(fuse_sum_relu_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, a, (float*)out);
    else hipLaunchKernelGGL(fuse_sum_relu_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, a, (bf16_t*)out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_blocksum_masked(const void* g, const void* y, void* out, int dtype, int B, int Hc, int Wc, int C, int shift,
                                     void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if ((C % 8) || shift < 0) return (int)hipErrorInvalidValue;
    const long nvec = (long)B * Hc * Wc * (C / 8);
    if (nvec <= 0) return 0;
    if (dtype == LOFT_F32)
        hipLaunchKernelGGL(blocksum_masked_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const float*)g,
                           (const float*)y, (float*)out, B, Hc, Wc, C, shift);
    else
        hipLaunchKernelGGL(blocksum_masked_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                           (const bf16_t*)y, (bf16_t*)out, B, Hc, Wc, C, shift);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_bilinear_up_slot(const void* src, void* dst, int dtype, int B, int h, int w, int C, int shift, int Ctot,
                                      int coff, int backward, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if ((C % 8) || (Ctot % 8) || (coff % 8) || shift < 0) return (int)hipErrorInvalidValue;
    if (!backward) {
        const long nvec = (long)B * (h << shift) * (w << shift) * (C / 8);
        if (nvec <= 0) return 0;
        if (dtype == LOFT_F32)
            hipLaunchKernelGGL(bilinear_up_slot_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                               (float*)dst, B, h, w, C, shift, Ctot, coff);
        else
            hipLaunchKernelGGL(bilinear_up_slot_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)src, (bf16_t*)dst, B, h, w, C, shift, Ctot, coff);
    } else {   // src = gradient of the slotted tensor [B,H,W,Ctot]; dst = gradient of the small map [B,h,w,C]
        const long nvec = (long)B * h * w * (C / 8);
        if (nvec <= 0) return 0;
        if (dtype == LOFT_F32)
            hipLaunchKernelGGL(bilinear_up_slot_bwd_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream,
                               (const float*)src, (float*)dst, B, h, w, C, shift, Ctot, coff);
        else
            hipLaunchKernelGGL(bilinear_up_slot_bwd_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)src, (bf16_t*)dst, B, h, w, C, shift, Ctot, coff);
    }
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_avgpool(const void* src, void* dst, int dtype, int B, int Ho, int Wo, int C, int shift, int backward,
                             int accumulate, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    if ((C % 8) || shift < 0) return (int)hipErrorInvalidValue;
    if (!backward) {
        const long nvec = (long)B * Ho * Wo * (C / 8);
        if (nvec <= 0) return 0;
        if (dtype == LOFT_F32)
            hipLaunchKernelGGL(avgpool_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                               (float*)dst, B, Ho, Wo, C, shift);
        else
            hipLaunchKernelGGL(avgpool_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                               (bf16_t*)dst, B, Ho, Wo, C, shift);
    } else {   // src = gradient of the pooled map [B,Ho,Wo,C]; dst = gradient of the fine map [B,Ho<<s,Wo<<s,C]
        const long nvec = (long)B * (Ho << shift) * (Wo << shift) * (C / 8);
        if (nvec <= 0) return 0;
        if (dtype == LOFT_F32)
            hipLaunchKernelGGL(avgpool_bwd_kernel<float>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                               (float*)dst, B, Ho, Wo, C, shift, accumulate);
        else
            hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, grid_for(nvec), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                               (bf16_t*)dst, B, Ho, Wo, C, shift, accumulate);
    }
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_stem3x3s2_bn_relu(const float* img, const float* w, const float* scale, const float* shift, void* out,
                                       int dtype, int B, int H, int W, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    dim3 grid(loft_cdiv(Wo, 16), loft_cdiv(Ho, 16), B);
    if (dtype == LOFT_F32)
        hipLaunchKernelGGL(stem3x3s2_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, w, scale, shift, (float*)out, B, H,
                           W, Ho, Wo);
    else
        hipLaunchKernelGGL(stem3x3s2_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, img, w, scale, shift, (bf16_t*)out, B,
                           H, W, Ho, Wo);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_stem3x3s2_wgrad(const float* img, const void* g, const void* y, float* dwp, float* db, int dtype, int B,
                                     int H, int W, void* stream) {
    if (dtype != LOFT_F32 && dtype != LOFT_ACT16) return (int)hipErrorInvalidValue;   // the other build's 16-bit type
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int tiles_x = loft_cdiv(Wo, 16), tiles_y = loft_cdiv(Ho, 16);
    int blocks = tiles_x * tiles_y * B;
    if (blocks > 1024) blocks = 1024;
    if (dtype == LOFT_F32)
        hipLaunchKernelGGL(stem3x3s2_wgrad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (const float*)g,
                           (const float*)y, dwp, db, B, H, W, Ho, Wo, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(stem3x3s2_wgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (const bf16_t*)g,
                           (const bf16_t*)y, dwp, db, B, H, W, Ho, Wo, tiles_x, tiles_y);
    LOFT_LAUNCH_CHECK();
    return 0;
}
