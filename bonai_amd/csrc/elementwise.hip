// elementwise.hip -- HBM-bound glue kernels of the LOFT path (gfx950), all NHWC, 16-byte vector accesses.
//
//   relu_bwd        : g * (y > 0)                    (autograd of F.relu / ConvModule activation)
//   colsum          : per-channel sum over pixels     (bias gradients; frozen-BN beta gradients)
//   upsample_add    : lat[l] += nearest_x2(lat[l+1])  (mmdet/models/necks/fpn.py:176-181) and its adjoint
//   subsample2      : P6 = max_pool2d(P5, 1, stride=2) (fpn.py:189-191) and its adjoint
//   maxpool3x3s2    : ResNet stem pooling              (mmdet/models/backbones/resnet.py:631)
//   stem7x7         : conv 7x7/2 (3->64) + frozen BN + ReLU, fp32 NCHW image -> bf16 NHWC (resnet.py:628-630)
//   cast / add      : fp32 accumulators -> bf16, bf16 a+b
//   sgd_momentum    : fused gradient-clip scale + weight-decay + momentum SGD on a flat fp32 arena
//                     (mmcv OptimizerHook(grad_clip) + torch.optim.SGD, schedule_2x_bonai.py:2-3)
// Roofline for all of these: HBM bandwidth.
#include "loft_common.h"
#include "../../include/loft_hip.h"
#include "conv_tap.h"      // planes_scale_of

__device__ __forceinline__ void ld8(const bf16_t* p, float v[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack8_16(t, v);
}
__device__ __forceinline__ void st8(bf16_t* p, const float v[8]) {
    *reinterpret_cast<uint4*>(p) = pack8_16(v);
}

// fp32 parity mode: the same 8-channel groups, as two 16-byte accesses
__device__ __forceinline__ void ld8(const float* p, float v[8]) { ld4(p, v); ld4(p + 4, v + 4); }
__device__ __forceinline__ void st8(float* p, const float v[8]) { st4(p, v); st4(p + 4, v + 4); }

static inline dim3 ew_grid(long nvec) {
    long b = (nvec + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// ---- relu backward -------------------------------------------------------------------------
__global__ void relu_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ y, bf16_t* __restrict__ out,
                                long nvec) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float gv[8], yv[8];
        ld8(g + i * 8, gv); ld8(y + i * 8, yv);
#pragma unroll
        for (int q = 0; q < 8; ++q) gv[q] = yv[q] > 0.f ? gv[q] : 0.f;
        st8(out + i * 8, gv);
    }
}
LOFT_EXPORT int loft_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(relu_bwd_kernel, ew_grid(n / 8), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                       (const bf16_t*)y, (bf16_t*)out, n / 8);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- column sum: out[c] += sum_m x[m][c]   (x bf16 [M][C], out fp32; caller zeroes) ----
// Each block reduces a slab of rows: thread t owns the 8-channel group (t % ncg) of rows t/ncg, t/ncg + rpar, ...
// (16-byte loads, a wave covers whole 128..512-byte rows), partial sums are combined through LDS and only
// the first row-group issues the C global atomics.  The slab height is chosen so that ~2k blocks exist
// whatever M is (layer4's M=8192 would otherwise leave 240 CUs idle).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, long M, int C, float* __restrict__ out,
                                                     int rows_per_block) {
    __shared__ float part[256 * 8];
    const int cg = C >> 3;
    const int ncg = cg < 256 ? cg : 256;   // channel groups handled per pass
    const int rpar = 256 / ncg;
    const int tc = threadIdx.x % ncg, tr = threadIdx.x / ncg;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int c8 = tc; c8 < cg; c8 += ncg) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tr < rpar)
            for (long r = r0 + tr; r < r1; r += rpar) {
                float v[8];
                ld8(x + r * C + c8 * 8, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += v[q];
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) part[threadIdx.x * 8 + q] = acc[q];
        __syncthreads();
        if (tr == 0) {
            for (int j = 1; j < rpar; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += part[(j * ncg + tc) * 8 + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) unsafeAtomicAdd(out + c8 * 8 + q, acc[q]);
        }
    }
}
LOFT_EXPORT int loft_colsum_bf16(const void* x, int64_t M, int C, float* out, void* stream) {
    if (M <= 0) return 0;
    if (C % 8) return (int)hipErrorInvalidValue;
    long rows = (M + 2047) / 2048;
    if (rows < 16) rows = 16;
    long blocks = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)M,
                       C, out, (int)rows);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- FPN top-down: fine[b,y,x,:] += coarse[b,y/2,x/2,:] ; adjoint: coarse += sum of the 2x2 block ----
template <typename T>
__global__ void upsample_add_kernel(T* __restrict__ fine, const T* __restrict__ coarse, int B, int H, int W, int C) {
    const long nvec = (long)B * H * W * (C >> 3);
    const int cg = C >> 3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        const long ci = (((long)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + c8 * 8;
        float a[8], c[8];
        ld8(fine + i * 8, a); ld8(coarse + ci, c);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += c[q];
        st8(fine + i * 8, a);
    }
}
LOFT_EXPORT int loft_upsample2x_add_bf16(void* fine, const void* coarse, int B, int H, int W, int C, void* stream) {
    if ((C % 8) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(upsample_add_kernel<bf16_t>, ew_grid((long)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)fine, (const bf16_t*)coarse, B, H, W, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_upsample2x_add_f32(float* fine, const float* coarse, int B, int H, int W, int C, void* stream) {
    if ((C % 8) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(upsample_add_kernel<float>, ew_grid((long)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, fine,
                       coarse, B, H, W, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// out = coarse + 2x2 block sums of fine; out may BE coarse (the in-place entry) -- every lane reads its own 8 values before it
// writes them
__global__ void downsum_add_kernel(bf16_t* out, const bf16_t* coarse, const bf16_t* __restrict__ fine, int B, int Hc, int Wc,
                                   int C) {
    const int cg = C >> 3;
    const long nvec = (long)B * Hc * Wc * cg;
    const int W = Wc * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % Wc); p /= Wc;
        const int y = (int)(p % Hc);
        const int b = (int)(p / Hc);
        const long f0 = (((long)b * Hc * 2 + y * 2) * W + x * 2) * C + c8 * 8;
        float a[8], t[8];
        ld8(coarse + i * 8, a);
        ld8(fine + f0, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += t[q];
        ld8(fine + f0 + C, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += t[q];
        ld8(fine + f0 + (long)W * C, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += t[q];
        ld8(fine + f0 + (long)W * C + C, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += t[q];
        st8(out + i * 8, a);
    }
}
LOFT_EXPORT int loft_downsum2x_add_bf16(void* coarse, const void* fine, int B, int Hc, int Wc, int C, void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(downsum_add_kernel, ew_grid((long)B * Hc * Wc * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)coarse, (const bf16_t*)coarse, (const bf16_t*)fine, B, Hc, Wc, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_downsum2x_sum_bf16(void* out, const void* coarse, const void* fine, int B, int Hc, int Wc, int C, void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(downsum_add_kernel, ew_grid((long)B * Hc * Wc * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)out, (const bf16_t*)coarse, (const bf16_t*)fine, B, Hc, Wc, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- stride-2 subsample (P6) and its adjoint (scatter-add into the even positions) ----
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                  int adjoint) {
    const int cg = C >> 3;
    const long nvec = (long)B * Ho * Wo * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const long big = (((long)b * Hi + y * 2) * Wi + x * 2) * C + c8 * 8;
        float v[8];
        if (!adjoint) {
            ld8(src + big, v);
            st8(dst + i * 8, v);
        } else {  // dst (big) += src (small)
            float a[8];
            ld8(src + i * 8, v); ld8(dst + big, a);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += v[q];
            st8(dst + big, a);
        }
    }
}
LOFT_EXPORT int loft_subsample2_bf16(const void* src, void* dst, int B, int Ho, int Wo, int Hi, int Wi, int C, int adjoint,
                                     void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(subsample2_kernel<bf16_t>, ew_grid((long)B * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, B, Ho, Wo, Hi, Wi, C, adjoint);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_subsample2_f32(const float* src, float* dst, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(subsample2_kernel<float>, ew_grid((long)B * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       B, Ho, Wo, Hi, Wi, C, 0);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- max pool 3x3 stride 2 pad 1 (forward only: it sits in the frozen stem) ----
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Hi, int Wi, int Ho, int Wo,
                                    int C) {
    const int cg = C >> 3;
    const long nvec = (long)B * Ho * Wo * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -3.0e38f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = y * 2 + dy;
            if (iy < 0 || iy >= Hi) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = x * 2 + dx;
                if (ix < 0 || ix >= Wi) continue;
                float v[8];
                ld8(src + (((long)b * Hi + iy) * Wi + ix) * C + c8 * 8, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], v[q]);
            }
        }
        st8(dst + i * 8, m);
    }
}
LOFT_EXPORT int loft_maxpool3x3s2_bf16(const void* src, void* dst, int B, int Hi, int Wi, int C, void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, ew_grid((long)B * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, B, Hi, Wi, Ho, Wo, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_maxpool3x3s2_f32(const float* src, float* dst, int B, int Hi, int Wi, int C, void* stream) {
    if (C % 8) return (int)hipErrorInvalidValue;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, ew_grid((long)B * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, B, Hi, Wi, Ho, Wo, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- stem: conv 7x7 stride 2 pad 3, 3 -> 64, + folded frozen BN (scale/shift) + ReLU ----
// img fp32 NCHW [B,3,H,W] (what the reference's data pipeline hands over); w fp32 [64][3][7][7];
// out bf16 NHWC [B,H/2,W/2,64].  Block = 16x16 output pixels; the 37x37x3 input patch and the
// 64x147 weights live in LDS; each thread produces one pixel x 64 channels in 4 passes of 16.
template <typename T>
__global__ __launch_bounds__(256) void stem7x7_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      T* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
    __shared__ float patch[3][37][38];
    __shared__ float wl[147][64];  // [k][oc]
    const int tid = threadIdx.x;
    for (int i = tid; i < 147 * 64; i += 256) {
        const int oc = i / 147, k = i - oc * 147;
        wl[k][oc] = w[i];
    }
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = tid; i < 3 * 37 * 37; i += 256) {
        const int c = i / (37 * 37), r = i - c * 37 * 37;
        const int py = r / 37, px = r - py * 37;
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
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const float v = patch[c][ty * 2 + r][tx * 2 + s];
                    const float* wk = &wl[(c * 7 + r) * 7 + s][pass * 16];
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
LOFT_EXPORT int loft_stem7x7_bn_relu(const float* img, const float* w, const float* scale, const float* shift, void* out,
                                     int B, int H, int W, int out_f32, void* stream) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid(loft_cdiv(Wo, 16), loft_cdiv(Ho, 16), B);
    if (out_f32)
        hipLaunchKernelGGL(stem7x7_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, w, scale, shift, (float*)out, B, H,
                           W, Ho, Wo);
    else
        hipLaunchKernelGGL(stem7x7_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, img, w, scale, shift, (bf16_t*)out, B,
                           H, W, Ho, Wo);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- casts / adds ----
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long nvec) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float v[8];
        float4 a = *reinterpret_cast<const float4*>(src + i * 8);
        float4 b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        st8(dst + i * 8, v);
    }
}
LOFT_EXPORT int loft_act16_dtype(void) { return LOFT_ACT16; }

LOFT_EXPORT int loft_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, ew_grid(n / 8), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 8);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- operand planes of the fp32 parity mode (loft_conv_tap_planes / loft_conv_wgrad_planes) ----
// An fp32 tensor x becomes NP tensors of the build's 16-bit type with x * scale = plane_0 + plane_1 (+ plane_2) to 2^-22 (binary16
// build: two planes of 11 significant bits, scale = the power of two that puts the tensor's absmax into [2^14, 2^15) -- binary16
// has five exponent bits) or 2^-24 (bfloat16 build: three planes of 8 bits, no scale: bfloat16 has fp32's exponent range):
// plane_k = RNE16(remainder), remainder -= plane_k (exact in fp32).  The contraction kernels then multiply planes pairwise on the
// 16-bit matrix cores and accumulate in fp32 -- Dekker / Ozaki-style splitting with the products kept to the order of the error.
#ifdef LOFT_ACT_F16
#define LOFT_PLANES 2
#else
#define LOFT_PLANES 3
#endif
__device__ __forceinline__ float absmax_block(const float* __restrict__ x, long nvec) {
    float m = 0.f;
    bool bad = false;
    // four independent 16-byte loads per thread and trip (round 6): with one, a 1024-workgroup grid kept 4 MB in flight and the pass
    // ran at ~1.3 TB/s (146 launches, 6.2 ms per fp32-mode step: profiles/round5_probes/fp32_planes_kernel_stats.csv)
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(x + (i + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a[u].x), fabsf(a[u].y)), fmaxf(fabsf(a[u].z), fabsf(a[u].w))));
            bad |= (a[u].x != a[u].x) | (a[u].y != a[u].y) | (a[u].z != a[u].z) | (a[u].w != a[u].w);
        }
    }
    for (; i < nvec; i += stride) {
        const float4 a = *reinterpret_cast<const float4*>(x + i * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        bad |= (a.x != a.x) | (a.y != a.y) | (a.z != a.z) | (a.w != a.w);
    }
    if (bad) m = __builtin_inff();                        // (NaN anywhere: fmaxf drops it -- report "no finite scale" instead)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // one atomic per WORKGROUP: same-address atomics serialise at ~10 ns each (one per wave of an 8192-block grid was 0.3 ms)
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* __restrict__ x, long nvec, unsigned* __restrict__ out) {
    const float m = absmax_block(x, nvec);
    if (threadIdx.x == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));  // (non-negative floats order as their bits)
}
// amax_out: a PRE-ZEROED word (the caller's slot pool: one memset per few thousand tensors instead of one per tensor)
LOFT_EXPORT int loft_absmax_f32(const float* x, int64_t n, float* amax_out, void* stream) {
    if (n % 4) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    dim3 grid = ew_grid(n / 4);
    if (grid.x > 1024) grid.x = 1024;                     // four workgroups per CU stream at the HBM rate; 1024 atomics at the end
    hipLaunchKernelGGL(absmax_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, n / 4, (unsigned*)amax_out);
    LOFT_LAUNCH_CHECK();
    return 0;
}
__device__ __forceinline__ void split_planes_body(const float* __restrict__ x, bf16_t* __restrict__ planes, long nvec, long n, float sc) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float r[8], h[8];
        const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        r[0] = a.x * sc; r[1] = a.y * sc; r[2] = a.z * sc; r[3] = a.w * sc; r[4] = b.x * sc; r[5] = b.y * sc; r[6] = b.z * sc; r[7] = b.w * sc;
#pragma unroll
        for (int p = 0; p < LOFT_PLANES; ++p) {
            const uint4 pk = pack8_16(r);
            *reinterpret_cast<uint4*>(planes + p * n + i * 8) = pk;
            if (p + 1 < LOFT_PLANES) {
                unpack8_16(pk, h);
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] -= h[q];
            }
        }
    }
}
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, bf16_t* __restrict__ planes, long nvec, long n,
                                                           const float* __restrict__ amax) {
    split_planes_body(x, planes, nvec, n, amax ? planes_scale_of(*amax, false) : 1.f);
}
// Small tensors (packed weights, RoI-level maps of a few MB): absmax AND split in ONE launch.  Every workgroup of the (<= 256
// workgroup, hence co-resident) grid publishes its maximum, arrives at a counter, waits for the others, reads the tensor's
// absmax and splits its share -- two launches and their boundary less per tensor, ~400 of them per step of the fp32 mode.
// slot[0] = absmax (float bits), slot[1] = arrival counter: both PRE-ZEROED by the caller; slot[0] holds the absmax afterwards.
__global__ __launch_bounds__(256) void absmax_split_fused_kernel(const float* __restrict__ x, bf16_t* __restrict__ planes, long n,
                                                                 unsigned* __restrict__ slot) {
    const float m = absmax_block(x, n / 4);
    __shared__ float amax_s;
    if (threadIdx.x == 0) {
        if (m > 0.f) atomicMax(slot, __float_as_uint(m));
        __threadfence();
        atomicAdd(slot + 1, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 26))
            __builtin_amdgcn_s_sleep(2);
        // (the grid is <= 256 workgroups of 256 threads -- co-resident on an idle chip; if other streams hold CUs for seconds the
        //  rendezvous cannot complete: abort the launch LOUDLY rather than split with a partial absmax -- a too-small scale would
        //  overflow the binary16 planes to inf without any error)
        if (spins >= (1u << 26)) __builtin_trap();
        __threadfence();
        amax_s = __uint_as_float(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    split_planes_body(x, planes, n / 8, n, planes_scale_of(amax_s, false));
}
LOFT_EXPORT int loft_planes_per_tensor(void) { return LOFT_PLANES; }
LOFT_EXPORT int loft_split_planes_f32(const float* x, int64_t n, void* planes, const float* amax, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(split_planes_kernel, ew_grid(n / 8), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)planes, n / 8, (long)n, amax);
    LOFT_LAUNCH_CHECK();
    return 0;
}
// absmax + split of one tensor: slot = two PRE-ZEROED 32-bit words (absmax, scratch); slot[0] is the absmax afterwards (the value
// the contraction kernels take as amax_*).  One fused launch up to 4 Mi elements, two launches above.
LOFT_EXPORT int loft_absmax_split_planes_f32(const float* x, int64_t n, void* planes, float* slot, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return (int)hipErrorInvalidValue;
#ifndef LOFT_ACT_F16
    if (slot == nullptr) return loft_split_planes_f32(x, n, planes, nullptr, stream);     // bfloat16 planes carry no scale
#endif
    if (slot == nullptr) return (int)hipErrorInvalidValue;
    if (n <= (4l << 20)) {
        const long nb = (n / 8 + 255) / 256;
        hipLaunchKernelGGL(absmax_split_fused_kernel, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16_t*)planes, (long)n, (unsigned*)slot);
        LOFT_LAUNCH_CHECK();
        return 0;
    }
    const int e = loft_absmax_f32(x, n, slot, stream);
    return e ? e : loft_split_planes_f32(x, n, planes, slot, stream);
}
// ---- fp32 parity mode under the trainer: BN fold + fp32 operand packings + their planes for EVERY registered conv in two launches
// per step (kernels.PrepackRegistry.request_f32).  Per conv and step the mode ran loft_fold_pack (13.8 us) and, for the forward and
// the data-gradient packing each, loft_absmax_split_planes_f32 (22.7 us: a grid rendezvous per tensor): 108 + 169 launches, 5.3 of
// 121 ms per step (profiles/round6_probes/fp32_profile.txt).  The weights only change in the SGD kernel.
// Launch 1 (fold): record = 16 int64 {w, conv_bias, gamma, beta, mean, var, wp_fwd, wp_dgrad, bias_out, eps (float bits), Cout, Cin,
//   RS, CoutP, CinP, first_block} + 2 int64 {amax slot, unused}; a block = 1024 elements of the padded forward packing; values and
//   operation order of fold_pack_kernel<float>; the record's absmax (forward and data-gradient packing hold the same values) is
//   folded into its PRE-ZEROED slot (members of one grouped launch share a slot: one scale per operand tensor).
// Launch 2 (split): record = 6 int64 {src fp32, planes dst, plane stride (elements), count, amax slot | 0, first_block}; a block =
//   2048 elements; split_planes_body's arithmetic.
__global__ __launch_bounds__(256) void fold_f32_multi_kernel(const long* __restrict__ desc, int n, long nblocks) {
    __shared__ float wm[4];
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        int lo;
        {
            int cnt = 0;
            for (int j = threadIdx.x & 63; j < n; j += 64) cnt += desc[(long)j * 18 + 15] <= blk ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            lo = cnt - 1;
        }
        const long* d = desc + (long)lo * 18;
        const float* w = reinterpret_cast<const float*>(d[0]);
        const float* cbias = reinterpret_cast<const float*>(d[1]);
        const float* gamma = reinterpret_cast<const float*>(d[2]);
        const float* beta = reinterpret_cast<const float*>(d[3]);
        const float* mean = reinterpret_cast<const float*>(d[4]);
        const float* var = reinterpret_cast<const float*>(d[5]);
        float* wp = reinterpret_cast<float*>(d[6]);
        float* wpt = reinterpret_cast<float*>(d[7]);
        float* bias_out = reinterpret_cast<float*>(d[8]);
        const float eps = __int_as_float((int)d[9]);
        const int Cout = (int)d[10], Cin = (int)d[11], RS = (int)d[12], CoutP = (int)d[13], CinP = (int)d[14];
        unsigned* slot = reinterpret_cast<unsigned*>(d[16]);
        const long total = (long)CoutP * CinP * RS, b0 = blk - d[15];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = b0 * 1024 + u * 256 + threadIdx.x;
            if (i < total) {
                const int c = (int)(i % CinP);
                const long r = i / CinP;
                const int nn = (int)(r % CoutP), t = (int)(r / CoutP);
                float v = 0.f;
                if (nn < Cout && c < Cin) {
                    v = w[((long)nn * Cin + c) * RS + t];
                    if (gamma) v *= gamma[nn] * rsqrtf(var[nn] + eps);
                }
                if (wp) wp[i] = v;
                if (wpt) wpt[((long)t * CinP + c) * CoutP + nn] = v;
                m = fmaxf(m, v != v ? __builtin_inff() : fabsf(v));
            }
        }
        if (bias_out && b0 == 0)
            for (int nn = threadIdx.x; nn < CoutP; nn += blockDim.x) {
                if (nn >= Cout) bias_out[nn] = 0.f;
                else if (gamma) bias_out[nn] = beta[nn] - mean[nn] * gamma[nn] * rsqrtf(var[nn] + eps);
                else bias_out[nn] = cbias ? cbias[nn] : 0.f;
            }
        if (slot) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            __syncthreads();
            if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                const float mm = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
                if (mm > 0.f) atomicMax(slot, __float_as_uint(mm));
            }
        }
    }
}
__global__ __launch_bounds__(256) void split_f32_multi_kernel(const long* __restrict__ desc, int n, long nblocks) {
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        int lo;
        {
            int cnt = 0;
            for (int j = threadIdx.x & 63; j < n; j += 64) cnt += desc[(long)j * 6 + 5] <= blk ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            lo = cnt - 1;
        }
        const long* d = desc + (long)lo * 6;
        const float* x = reinterpret_cast<const float*>(d[0]);
        bf16_t* planes = reinterpret_cast<bf16_t*>(d[1]);
        const long stride = d[2], count = d[3];
        const float* amax = reinterpret_cast<const float*>(d[4]);
        const float sc = amax ? planes_scale_of(*amax, false) : 1.f;
        const long i = (blk - d[5]) * 256 + threadIdx.x;              // 8 elements per thread
        if (i * 8 < count) {
            float r[8], h[8];
            const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
            const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
            r[0] = a.x * sc; r[1] = a.y * sc; r[2] = a.z * sc; r[3] = a.w * sc; r[4] = b.x * sc; r[5] = b.y * sc; r[6] = b.z * sc; r[7] = b.w * sc;
#pragma unroll
            for (int p = 0; p < LOFT_PLANES; ++p) {
                const uint4 pk = pack8_16(r);
                *reinterpret_cast<uint4*>(planes + p * stride + i * 8) = pk;
                if (p + 1 < LOFT_PLANES) {
                    unpack8_16(pk, h);
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] -= h[q];
                }
            }
        }
    }
}
LOFT_EXPORT int loft_fold_f32_multi(const int64_t* desc, int n, int64_t nblocks, void* stream) {
    if (n <= 0 || nblocks <= 0) return 0;
    const long g = nblocks < 8192 ? nblocks : 8192;
    hipLaunchKernelGGL(fold_f32_multi_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const long*)desc, n, (long)nblocks);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_split_planes_f32_multi(const int64_t* desc, int n, int64_t nblocks, void* stream) {
    if (n <= 0 || nblocks <= 0) return 0;
    const long g = nblocks < 16384 ? nblocks : 16384;
    hipLaunchKernelGGL(split_f32_multi_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const long*)desc, n, (long)nblocks);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// out[g][c] += sum over the rows of x[g][rows][C]: the bias gradient of the fp32 mode's convolutions (a strided library reduction
// over the NHWC gradient took 0.44 ms per layer).  A workgroup = 4 row phases x 64 lanes of 4 channels (C <= 256 per pass), its
// partial sums meet in LDS and leave as one atomic per channel.  out accumulates (the caller zeroes it).  C % 4 == 0.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ out) {
    const float* xg = x + (long)blockIdx.y * rows * C;
    float* og = out + (long)blockIdx.y * C;
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    __shared__ float part[4][256];
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C)
            for (long r = (long)blockIdx.x * 4 + ph; r < rows; r += (long)gridDim.x * 4) {
                const float4 v = *reinterpret_cast<const float4*>(xg + r * C + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        part[ph][lane * 4 + 0] = acc.x; part[ph][lane * 4 + 1] = acc.y; part[ph][lane * 4 + 2] = acc.z; part[ph][lane * 4 + 3] = acc.w;
        __syncthreads();
        const int cc = c0 + threadIdx.x;
        if (cc < C) unsafeAtomicAdd(og + cc, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
        __syncthreads();
    }
}
LOFT_EXPORT int loft_colsum_f32(const float* x, int64_t rows, int C, int groups, float* out, void* stream) {
    if (rows <= 0 || groups < 1) return 0;
    if (C % 4 || C < 4) return (int)hipErrorInvalidValue;
    long nb = (rows + 63) / 64;
    nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)nb, groups), dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

__global__ void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out,
                                long nvec) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float x[8], y[8];
        ld8(a + i * 8, x); ld8(b + i * 8, y);
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] += y[q];
        st8(out + i * 8, x);
    }
}
LOFT_EXPORT int loft_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(add_bf16_kernel, ew_grid(n / 8), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)out, n / 8);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- optimizer: sum of squares (for clip_grad_norm_) and fused SGD step on a flat fp32 arena ----
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
    // 16-byte loads, four independent chains per lane (a pure HBM stream: 4-byte loads reached 2.2 TB/s)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n >> 2 : 0;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four loads in flight per lane
        const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
        a0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        a1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        a2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
        a3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        a0 += v.x * v.x; a1 += v.y * v.y; a2 += v.z * v.z; a3 += v.w * v.w;
    }
    for (long j = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += stride) {
        const float v = g[j];
        a0 += v * v;
    }
    float acc = (a0 + a1) + (a2 + a3);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
LOFT_EXPORT int loft_sumsq_f32(const float* g, int64_t n, float* out, void* stream) {
    if (n <= 0) return 0;
    // <= 1024 workgroups: every workgroup ends in ONE atomic on the same scalar, and 8192 of those serialised in L2 for ~80 us of the
    // launch's 121 (a 166 MB read is 35 us of HBM time); 1024 x 256 lanes x 4 x 16 B = 16 MB in flight still saturates the memory
    dim3 grid = ew_grid(n / 16);
    if (grid.x > 1024u) grid.x = 1024u;
    hipLaunchKernelGGL(sumsq_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, (long)n, out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// p, g, m: flat fp32 [n].  gnorm_sq: device scalar (sum of squares of ALL grads, after all-reduce
// averaging).  clip = max_norm / (norm + 1e-6) if norm > max_norm else 1 (torch clip_grad_norm_).
// torch.optim.SGD: d = g*clip + wd*p ; m = mu*m + d ; p -= lr*m   (first step m = d is the caller's
// business: start from m = 0 and it is identical).
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n,
                           const float* __restrict__ gnorm_sq, float max_norm, float lr, float mu, float wd,
                           float gscale) {
    float clip = 1.f;
    if (max_norm > 0.f) {
        const float norm = sqrtf(*gnorm_sq) * gscale;
        if (norm > max_norm) clip = max_norm / (norm + 1e-6f);
    }
    const float s = clip * gscale;
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    // 16-byte accesses on the aligned body (five HBM streams), scalar tail
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m)) & 15) == 0;
    const long n4 = al ? n >> 2 : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    auto upd = [&](float4& pv, float4& mv, const float4 gv) {
        mv.x = mu * mv.x + (gv.x * s + wd * pv.x); pv.x -= lr * mv.x;
        mv.y = mu * mv.y + (gv.y * s + wd * pv.y); pv.y -= lr * mv.y;
        mv.z = mu * mv.z + (gv.z * s + wd * pv.z); pv.z -= lr * mv.z;
        mv.w = mu * mv.w + (gv.w * s + wd * pv.w); pv.w -= lr * mv.w;
    };
    long i = t0;
    for (; i + stride < n4; i += 2 * stride) {              // six loads in flight per lane (three gave 2.7 TB/s over the five streams)
        float4 pa = p4[i], ma = m4[i], pb = p4[i + stride], mb = m4[i + stride];
        const float4 ga = g4[i], gb = g4[i + stride];
        upd(pa, ma, ga);
        upd(pb, mb, gb);
        m4[i] = ma; p4[i] = pa;
        m4[i + stride] = mb; p4[i + stride] = pb;
    }
    for (; i < n4; i += stride) {
        float4 pv = p4[i], mv = m4[i];
        upd(pv, mv, g4[i]);
        m4[i] = mv; p4[i] = pv;
    }
    for (long i = n4 * 4 + t0; i < n; i += stride) {
        const float pv = p[i];
        const float d = g[i] * s + wd * pv;
        const float mv = mu * m[i] + d;
        m[i] = mv;
        p[i] = pv - lr * mv;
    }
}
LOFT_EXPORT int loft_sgd_momentum_f32(float* p, const float* g, float* m, int64_t n, const float* gnorm_sq, float max_norm,
                                      float lr, float momentum, float weight_decay, float grad_scale, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, ew_grid(n / 4), dim3(256), 0, (hipStream_t)stream, p, g, m, (long)n, gnorm_sq, max_norm, lr,
                       momentum, weight_decay, grad_scale);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---- weight fold + pack (one launch per conv per step) and its chain rule -------------------------------
// Forward: from the fp32 master weight in the reference layout [Cout][Cin][R][S] (+ optional frozen-statistics BN
// gamma/beta/mean/var, tools/fuse_conv_bn.py:10-23) produce in ONE pass the bf16 operand packings the MFMA kernels
// consume: fwd [R*S][Cout][Cin], dgrad [R*S][Cin][Cout] (either may be NULL), and the fp32 epilogue bias
// (BN shift, or the conv bias).  Backward: from the packed fp32 weight gradient of the folded weight and the bias
// gradient produce dW in the reference layout, dgamma, dbeta.
template <typename OT> __device__ __forceinline__ OT fold_cvt(float v);
template <> __device__ __forceinline__ bf16_t fold_cvt<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ float fold_cvt<float>(float v) { return v; }
template <typename OT>
__global__ void fold_pack_kernel(const float* __restrict__ w, const float* __restrict__ cbias, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
                                 float eps, int Cout, int Cin, int RS, OT* __restrict__ wp, OT* __restrict__ wpt,
                                 float* __restrict__ bias_out, int CoutP, int CinP) {
    const long total = (long)CoutP * CinP * RS;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the (channel-padded) fwd packing [t][n][c] (coalesced writes); gather from [n][c][t]
        const int c = (int)(i % CinP);
        const long r = i / CinP;
        const int n = (int)(r % CoutP), t = (int)(r / CoutP);
        float v = 0.f;
        if (n < Cout && c < Cin) {
            v = w[((long)n * Cin + c) * RS + t];
            if (gamma) v *= gamma[n] * rsqrtf(var[n] + eps);
        }
        const OT h = fold_cvt<OT>(v);
        if (wp) wp[i] = h;
        if (wpt) wpt[((long)t * CinP + c) * CoutP + n] = h;
    }
    if (bias_out && blockIdx.x == 0)
        for (int n = threadIdx.x; n < CoutP; n += blockDim.x) {
            if (n >= Cout) bias_out[n] = 0.f;
            else if (gamma) bias_out[n] = beta[n] - mean[n] * gamma[n] * rsqrtf(var[n] + eps);
            else bias_out[n] = cbias ? cbias[n] : 0.f;
        }
}
LOFT_EXPORT int loft_fold_pack(const float* w, const float* conv_bias, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, int Cout, int Cin, int RS, void* wp_fwd, void* wp_dgrad,
                               float* bias_out, int pack_f32, int CoutP, int CinP, void* stream) {
    if (CoutP < Cout || CinP < Cin) return (int)hipErrorInvalidValue;
    const long total = (long)CoutP * CinP * RS;
    if (total <= 0) return 0;
    if (pack_f32)
        hipLaunchKernelGGL(fold_pack_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, conv_bias, gamma, beta,
                           mean, var, eps, Cout, Cin, RS, (float*)wp_fwd, (float*)wp_dgrad, bias_out, CoutP, CinP);
    else
        hipLaunchKernelGGL(fold_pack_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, conv_bias, gamma, beta,
                           mean, var, eps, Cout, Cin, RS, (bf16_t*)wp_fwd, (bf16_t*)wp_dgrad, bias_out, CoutP, CinP);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// Record lookup of the batched kernels: largest j with desc[j][15] <= key (first_chunk / first_block are ascending).  A binary
// search is a chain of ~7 dependent global loads (~8 us before a short block does any work); here every lane tests its own
// entries with independent loads and the wave counts the hits.
__device__ __forceinline__ int find_record(const long* __restrict__ desc, int n, long key) {
    int cnt = 0;
    for (int j = threadIdx.x & 63; j < n; j += 64) cnt += desc[(long)j * 16 + 15] <= key ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    return cnt - 1;
}

// Batched form: ONE launch packs every registered conv of the model (the trainer knows the full list after the first step;
// weights only change in the SGD kernel, so all packings of a step can be produced up front).  desc: n records of 16 int64
// {w, conv_bias, gamma, beta, mean, var, wp_fwd, wp_dgrad, bias_out, eps (float bits), Cout, Cin, RS, CoutP, CinP, first_chunk};
// chunk c belongs to the record with the largest first_chunk <= c.  A chunk is one [NT output channels] x [64 input channels]
// tile with all RS taps (NT = 64 for RS == 1, else 32; RS <= FOLD_TILE_MAX_RS): the fp32 source rows are read coalesced
// (64*RS contiguous floats per output channel) into LDS, then written out as 128-byte runs of wp_fwd [T][CoutP][CinP] and
// 2*NT-byte runs of wp_dgrad [T][CinP][CoutP] -- the per-element form of loft_fold_pack scatters 2-byte stores for the
// transposed packing.  Records with more taps use chunks of FOLD_CHUNK elements of the per-element form.
constexpr int FOLD_CHUNK = 2048;
constexpr int FOLD_TILE_MAX_RS = 9;
constexpr int FOLD_NT_TAPS = 32, FOLD_NT_TAPS_LOG2 = 5;                 // output channels per tile for 1 < RS <= FOLD_TILE_MAX_RS
constexpr int FOLD_TILE_FLOATS = FOLD_NT_TAPS * (64 * FOLD_TILE_MAX_RS + 4) / 2;   // the bf16 tile; >= 64 * 68 / 2 (the RS == 1 tile)
__global__ __launch_bounds__(256) void fold_pack_multi_kernel(const long* __restrict__ desc, int n, long nchunks) {
    __shared__ float tile[FOLD_TILE_FLOATS];
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int lo = find_record(desc, n, c);                  // (block-uniform)
        const long* d = desc + (long)lo * 16;
        const float* w = reinterpret_cast<const float*>(d[0]);
        const float* cbias = reinterpret_cast<const float*>(d[1]);
        const float* gamma = reinterpret_cast<const float*>(d[2]);
        const float* beta = reinterpret_cast<const float*>(d[3]);
        const float* mean = reinterpret_cast<const float*>(d[4]);
        const float* var = reinterpret_cast<const float*>(d[5]);
        bf16_t* wp = reinterpret_cast<bf16_t*>(d[6]);
        bf16_t* wpt = reinterpret_cast<bf16_t*>(d[7]);
        float* bias_out = reinterpret_cast<float*>(d[8]);
        const float eps = __int_as_float((int)d[9]);
        const int Cout = (int)d[10], Cin = (int)d[11], CoutP = (int)d[13], CinP = (int)d[14];
        // RS < 0: "n-major" forward packing wp[n][t][c] (a Linear applied to a flattened [C,H,W] map: the K axis of the GEMM is
        // (t, c) = the NHWC order of the activation, so the weight's (c, t) columns are permuted here instead of the activation)
        const bool nmajor = d[12] < 0;
        const int RS = (int)(nmajor ? -d[12] : d[12]);
        const int local_chunk = (int)(c - d[15]);
        if (local_chunk == 0 && bias_out)
            for (int nn = threadIdx.x; nn < CoutP; nn += blockDim.x) {
                if (nn >= Cout) bias_out[nn] = 0.f;
                else if (gamma) bias_out[nn] = beta[nn] - mean[nn] * gamma[nn] * rsqrtf(var[nn] + eps);
                else bias_out[nn] = cbias ? cbias[nn] : 0.f;
            }
        if (nmajor) {
            // one output channel per chunk: the fp32 row w[n][c][t] is read contiguously, permuted to (t, c) order through LDS
            // (as bf16: C * RS * 2 bytes <= the tile array) and written contiguously.  The transposed packing of these
            // records is produced afterwards by loft_transpose_bf16 (a [O][K] -> [K][O] tiled transpose).
            bf16_t* row = reinterpret_cast<bf16_t*>(tile);
            const int nn = local_chunk, per = Cin * RS;
            const float sc = gamma ? gamma[nn] * rsqrtf(var[nn] + eps) : 1.f;
            __syncthreads();
            // (j / RS by multiply-high: exact for j < 2^16 when per < 2^16 -- a runtime integer division is ~40 instructions per
            //  element, and the two 12544 x 1024 FC records are 25.7 M elements per step)
            const unsigned rs_mul = RS == 1 ? 0u : 0xFFFFFFFFu / (unsigned)RS + 1u;
            const bool small = per < 65536;
            for (int j = threadIdx.x; j < per; j += 256) {
                const int cc = RS == 1 ? j : (small ? (int)__umulhi((unsigned)j, rs_mul) : j / RS), t = j - cc * RS;
                row[t * Cin + cc] = f32_to_bf16(w[(long)nn * per + j] * sc);
            }
            __syncthreads();
            if (wp)
                for (int j = threadIdx.x * 2; j < per; j += 512)
                    *reinterpret_cast<uint32_t*>(wp + (long)nn * per + j) = *reinterpret_cast<const uint32_t*>(row + j);
            continue;
        }
        if (RS > FOLD_TILE_MAX_RS) {                             // per-element form
            const long total = (long)CoutP * CinP * RS;
            const long i0 = (long)local_chunk * FOLD_CHUNK;
            for (long i = i0 + threadIdx.x; i < i0 + FOLD_CHUNK && i < total; i += blockDim.x) {
                const int cc = (int)(i % CinP);
                const long r = i / CinP;
                const int nn = (int)(r % CoutP), t = (int)(r / CoutP);
                float v = 0.f;
                if (nn < Cout && cc < Cin) {
                    v = w[((long)nn * Cin + cc) * RS + t];
                    if (gamma) v *= gamma[nn] * rsqrtf(var[nn] + eps);
                }
                const bf16_t h = f32_to_bf16(v);
                if (wp) wp[i] = h;
                if (wpt) wpt[((long)t * CinP + cc) * CoutP + nn] = h;
            }
            continue;
        }
        // ---- tiled records.  The tile is held as bf16 (folded + rounded once, on the way in): NT = 32 output channels for taps > 1,
        // so the transposed packing is written in 64-byte runs (16 channels gave 32-byte runs: half-used memory bursts on a
        // quarter of the launch's traffic); aligned records move 16 bytes per lane on all three sides (4-byte accesses left the
        // launch at 2.1-2.5 TB/s, instruction-bound).
        bf16_t* bt = reinterpret_cast<bf16_t*>(tile);
        const int NT = RS == 1 ? 64 : FOLD_NT_TAPS;
        const int tc = (CinP + 63) >> 6;
        const int n0 = (local_chunk / tc) * NT, c0 = (local_chunk % tc) * 64;
        const int span = 64 * RS, ldw = span + 4;                // (row pitch in bf16: 8-byte aligned rows, odd multiple of 2 words)
        __syncthreads();                                         // the previous chunk's readers are done with the tile
        const bool vec = (Cin & 63) == 0 && (CinP & 7) == 0 && (CoutP & 7) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
        // (index math without per-element integer division by run-time values -- reciprocal multiplies and shifts)
        const unsigned rs_mul = RS == 1 ? 0u : 0xFFFFFFFFu / (unsigned)RS + 1u;      // x / RS == umulhi(x, rs_mul) for x < 2^16
        if (vec) {
            const int q4 = span >> 2;                            // float4s per row
            const unsigned q_mul = 0xFFFFFFFFu / (unsigned)q4 + 1u;
            for (int i = threadIdx.x; i < NT * q4; i += 256) {
                const int row = (int)__umulhi((unsigned)i, q_mul), off = (i - row * q4) << 2;
                const int nn = n0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nn < Cout) {
                    v = *reinterpret_cast<const float4*>(w + ((long)nn * Cin + c0) * RS + off);
                    if (gamma) { const float sc = gamma[nn] * rsqrtf(var[nn] + eps); v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                }
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                *reinterpret_cast<uint2*>(bt + row * ldw + off) = pk;
            }
        } else {
            const unsigned sp_mul = 0xFFFFFFFFu / (unsigned)span + 1u;               // x / span likewise
            for (int i = threadIdx.x; i < NT * span; i += 256) {
                const int row = (int)__umulhi((unsigned)i, sp_mul), off = i - row * span;
                const int nn = n0 + row, cc = c0 + (RS == 1 ? off : (int)__umulhi((unsigned)off, rs_mul));
                float v = 0.f;
                if (nn < Cout && cc < Cin) {
                    v = w[((long)nn * Cin + c0) * RS + off];
                    if (gamma) v *= gamma[nn] * rsqrtf(var[nn] + eps);
                }
                bt[row * ldw + off] = f32_to_bf16(v);
            }
        }
        __syncthreads();
        const int ntsh = RS == 1 ? 6 : FOLD_NT_TAPS_LOG2;            // log2(NT)
        if (wp && vec) {                                         // [t][nn][8 cc]: 8 lanes = one 128-byte run
            for (int i = threadIdx.x; i < RS * NT * 8; i += 256) {
                const int l8 = i & 7, r2 = i >> 3, row = r2 & (NT - 1), t = r2 >> ntsh;
                const int nn = n0 + row;
                if (nn < CoutP) {
                    const bf16_t* tp = bt + row * ldw + (8 * l8) * RS + t;
                    uint32_t q[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = (uint32_t)tp[(2 * k) * RS] | ((uint32_t)tp[(2 * k + 1) * RS] << 16);
                    *reinterpret_cast<uint4*>(wp + ((long)t * CoutP + nn) * CinP + c0 + 8 * l8) = make_uint4(q[0], q[1], q[2], q[3]);
                }
            }
        } else if (wp) {                                         // [t][nn][cc pair]: 32 lanes = one 128-byte run
            for (int i = threadIdx.x; i < RS * NT * 32; i += 256) {
                const int cp = i & 31, r2 = i >> 5, row = r2 & (NT - 1), t = r2 >> ntsh;
                const int nn = n0 + row, cc = c0 + 2 * cp;
                if (nn < CoutP && cc < CinP) {
                    const bf16_t* tp = bt + row * ldw + 2 * cp * RS + t;
                    *reinterpret_cast<uint32_t*>(wp + ((long)t * CoutP + nn) * CinP + cc) = (uint32_t)tp[0] | ((uint32_t)tp[RS] << 16);
                }
            }
        }
        if (wpt && vec) {                                        // [t][cc][8 nn]: NT/8 lanes = one 2*NT-byte run
            const int gsh = ntsh - 3, gp = 1 << gsh;
            for (int i = threadIdx.x; i < RS * 64 * gp; i += 256) {
                const int g8 = i & (gp - 1), r2 = i >> gsh, ccl = r2 & 63, t = r2 >> 6;
                const int nn = n0 + 8 * g8;
                if (nn < CoutP) {
                    const bf16_t* tp = bt + (8 * g8) * ldw + ccl * RS + t;
                    uint32_t q[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = (uint32_t)tp[(2 * k) * ldw] | ((uint32_t)tp[(2 * k + 1) * ldw] << 16);
                    *reinterpret_cast<uint4*>(wpt + ((long)t * CinP + c0 + ccl) * CoutP + nn) = make_uint4(q[0], q[1], q[2], q[3]);
                }
            }
        } else if (wpt) {                                        // [t][cc][nn pair]: NT/2 lanes = one 2*NT-byte run
            const int hsh = ntsh - 1, hp = 1 << hsh;
            for (int i = threadIdx.x; i < RS * 64 * hp; i += 256) {
                const int np = i & (hp - 1), r2 = i >> hsh, ccl = r2 & 63, t = r2 >> 6;
                const int nn = n0 + 2 * np, cc = c0 + ccl;
                if (nn < CoutP && cc < CinP) {
                    const bf16_t* tp = bt + 2 * np * ldw + ccl * RS + t;
                    *reinterpret_cast<uint32_t*>(wpt + ((long)t * CinP + cc) * CoutP + nn) = (uint32_t)tp[0] | ((uint32_t)tp[ldw] << 16);
                }
            }
        }
    }
}
LOFT_EXPORT int loft_fold_pack_multi(const int64_t* desc, int n, int64_t nchunks, void* stream) {
    if (n <= 0 || nchunks <= 0) return 0;
    long blocks = nchunks < 16384 ? nchunks : 16384;
    hipLaunchKernelGGL(fold_pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const long*)desc, n,
                       (long)nchunks);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// dst[c][r] = src[r][c] for a row-major bf16 matrix [R][Cc] (64 x 64 tiles through LDS, both sides coalesced)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int Cc) {
    __shared__ bf16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int r = i >> 5, cp = (i & 31) * 2;
        uint32_t v = 0;
        if (r0 + r < R && c0 + cp < Cc) v = *reinterpret_cast<const uint32_t*>(src + (long)(r0 + r) * Cc + c0 + cp);
        t[r][cp] = (bf16_t)(v & 0xffff);
        t[r][cp + 1] = (bf16_t)(v >> 16);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int c = i >> 5, rp = (i & 31) * 2;
        if (c0 + c < Cc && r0 + rp < R)
            *reinterpret_cast<uint32_t*>(dst + (long)(c0 + c) * R + r0 + rp) = (uint32_t)t[rp][c] | ((uint32_t)t[rp + 1][c] << 16);
    }
}
LOFT_EXPORT int loft_transpose_bf16(const void* src, void* dst, int R, int Cc, void* stream) {
    if (R <= 0 || Cc <= 0) return 0;
    if ((R & 1) || (Cc & 1)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(loft_cdiv(Cc, 64), loft_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, R, Cc);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// one block per output channel n
__global__ __launch_bounds__(256) void fold_unpack_bwd_kernel(const float* __restrict__ dwp, const float* __restrict__ db,
                                                              const float* __restrict__ w, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ var,
                                                              float eps, int Cout, int Cin, int RS, float* __restrict__ dw,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int CoutP,
                                                              int CinP, int accumulate) {
    const int n = blockIdx.x;
    const float rs = gamma ? rsqrtf(var[n] + eps) : 1.f;
    const float scale = gamma ? gamma[n] * rs : 1.f;
    float acc = 0.f;
    const int per = Cin * RS;
    for (int j = threadIdx.x; j < per; j += blockDim.x) {
        const int c = j / RS, t = j - c * RS;       // j indexes dw[n][c][t] (coalesced writes)
        const float g = dwp[((long)t * CoutP + n) * CinP + c];
        const long wi = (long)n * per + j;
        if (dw) dw[wi] = (accumulate ? dw[wi] : 0.f) + g * scale;
        if (gamma) acc += g * w[wi];
    }
    if (gamma) {
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float s = part[0] + part[1] + part[2] + part[3];
            const float dbn = db ? db[n] : 0.f;
            if (dgamma) dgamma[n] = (accumulate ? dgamma[n] : 0.f) + (s * rs - dbn * mean[n] * rs);
            if (dbeta) dbeta[n] = (accumulate ? dbeta[n] : 0.f) + dbn;
        }
    }
}
// Batched form for the trainer's direct gradient sink: n records of 16 int64 {dwp, db, w, gamma, mean, var, dw, dgamma,
// dbeta_or_dbias, eps (float bits), Cout, Cin, RS, CoutP, CinP, first_block}; block b serves output channel
// b - first_block of the record with the largest first_block <= b and ACCUMULATES into dw / dgamma / dbeta (slots of the
// flat gradient arena).  Without BN (gamma == 0) the last slot is the conv's bias gradient: dbias[n] += db[n].
__global__ __launch_bounds__(256) void fold_unpack_bwd_multi_kernel(const long* __restrict__ desc, int njobs, int lds_floats) {
    const int lo = find_record(desc, njobs, (long)blockIdx.x);
    const long* d = desc + (long)lo * 16;
    const float* dwp = reinterpret_cast<const float*>(d[0]);
    const float* db = reinterpret_cast<const float*>(d[1]);
    const float* w = reinterpret_cast<const float*>(d[2]);
    const float* gamma = reinterpret_cast<const float*>(d[3]);
    const float* mean = reinterpret_cast<const float*>(d[4]);
    const float* var = reinterpret_cast<const float*>(d[5]);
    float* dw = reinterpret_cast<float*>(d[6]);
    float* dgamma = reinterpret_cast<float*>(d[7]);
    float* dbeta = reinterpret_cast<float*>(d[8]);
    const float eps = __int_as_float((int)(d[9] & 0xffffffffl));
    const int nsp = (int)(d[9] >> 32) > 1 ? (int)(d[9] >> 32) : 1;   // split-K slots of dwp to sum (loft_conv_wgrad_bf16_slots), else 1
    const int Cin = (int)d[11], CoutP = (int)d[13], CinP = (int)d[14];
    const bool nmajor = d[12] < 0;             // dwp is [n][t][c] (weight gradient of a Linear over an NHWC-flattened map)
    const int RS = (int)(nmajor ? -d[12] : d[12]);
    const int n = (int)(blockIdx.x - d[15]);
    const long sps = (long)RS * CoutP * CinP;  // elements per slot
    // (four slots per round: independent loads in flight instead of one load-add chain per slot)
    auto ld1 = [&](const float* p) {
        float v = *p;
        int q = 1;
        for (; q + 3 < nsp; q += 4) {
            const float u0 = p[q * sps], u1 = p[(q + 1) * sps], u2 = p[(q + 2) * sps], u3 = p[(q + 3) * sps];
            v += (u0 + u1) + (u2 + u3);
        }
        for (; q < nsp; ++q) v += p[q * sps];
        return v;
    };
    auto ld4s = [&](const float* p) {
        float4 v = *reinterpret_cast<const float4*>(p);
        int q = 1;
        for (; q + 3 < nsp; q += 4) {
            const float4 u0 = *reinterpret_cast<const float4*>(p + q * sps), u1 = *reinterpret_cast<const float4*>(p + (q + 1) * sps);
            const float4 u2 = *reinterpret_cast<const float4*>(p + (q + 2) * sps), u3 = *reinterpret_cast<const float4*>(p + (q + 3) * sps);
            v.x += (u0.x + u1.x) + (u2.x + u3.x); v.y += (u0.y + u1.y) + (u2.y + u3.y);
            v.z += (u0.z + u1.z) + (u2.z + u3.z); v.w += (u0.w + u1.w) + (u2.w + u3.w);
        }
        for (; q < nsp; ++q) {
            const float4 u = *reinterpret_cast<const float4*>(p + q * sps);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        return v;
    };
    const float rs = gamma ? rsqrtf(var[n] + eps) : 1.f;
    const float scale = gamma ? gamma[n] * rs : 1.f;
    float acc = 0.f;
    const int per = Cin * RS;
    if (nmajor) {
        // dwp row [t][c] -> the parameter's (c, t) order through LDS, in blocks of NMAJOR_CB channels: RS segments of CB floats in,
        // CB * RS contiguous floats out.  (The whole row at once -- 12 544 floats for the two 7 x 7 x 256 FC layers -- made the
        // launch's dynamic LDS 50 KB for EVERY block of the batch, the 2 304-float conv records included: 3 blocks per CU.)
        extern __shared__ float urow[];
        const int CB = min(Cin, lds_floats / RS);
        const unsigned cb_mul = CB == 1 ? 0u : 0xFFFFFFFFu / (unsigned)CB + 1u;        // j / CB by multiply-high (j < 2^16)
        for (int cb0 = 0; cb0 < Cin; cb0 += CB) {
            const int cbn = min(CB, Cin - cb0), cnt = cbn * RS;
            const bool small = CB * RS < 65536 && cbn == CB;
            __syncthreads();
            for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
                const int t = cbn == 1 ? j : (small ? (int)__umulhi((unsigned)j, cb_mul) : j / cbn), c = j - t * cbn;
                urow[c * RS + t] = ld1(dwp + (long)n * per + (long)t * CinP + cb0 + c);
            }
            __syncthreads();
            for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
                const long wi = (long)n * per + (long)cb0 * RS + j;
                if (dw) dw[wi] += urow[j] * scale;
            }
        }
        if (threadIdx.x == 0 && dbeta && db) dbeta[n] += db[n];
        return;
    }
    if (RS > 1 && per <= lds_floats) {
        // taps > 1: the RS planes dwp[t][n][:] are read contiguously and interleaved to the parameter's (c, t) order through LDS
        // (the direct form below reads 4 bytes from RS different planes per lane group: uncoalesced)
        extern __shared__ float urow[];
        if ((Cin & 3) == 0 && (CinP & 3) == 0) {          // 16-byte accesses on both sides
            for (int j = threadIdx.x * 4; j < per; j += blockDim.x * 4) {
                const int t = j / Cin, c = j - t * Cin;
                const float4 v = ld4s(dwp + ((long)t * CoutP + n) * CinP + c);
                urow[c * RS + t] = v.x; urow[(c + 1) * RS + t] = v.y; urow[(c + 2) * RS + t] = v.z; urow[(c + 3) * RS + t] = v.w;
            }
            __syncthreads();
            for (int j = threadIdx.x * 4; j < per; j += blockDim.x * 4) {
                const float4 g = *reinterpret_cast<const float4*>(urow + j);
                const long wi = (long)n * per + j;
                if (dw) {
                    float4 o = *reinterpret_cast<float4*>(dw + wi);
                    o.x += g.x * scale; o.y += g.y * scale; o.z += g.z * scale; o.w += g.w * scale;
                    *reinterpret_cast<float4*>(dw + wi) = o;
                }
                if (gamma) {
                    const float4 ww = *reinterpret_cast<const float4*>(w + wi);
                    acc += g.x * ww.x + g.y * ww.y + g.z * ww.z + g.w * ww.w;
                }
            }
        } else {
            for (int j = threadIdx.x; j < per; j += blockDim.x) {
                const int t = j / Cin, c = j - t * Cin;
                urow[c * RS + t] = ld1(dwp + ((long)t * CoutP + n) * CinP + c);
            }
            __syncthreads();
            for (int j = threadIdx.x; j < per; j += blockDim.x) {
                const float g = urow[j];
                const long wi = (long)n * per + j;
                if (dw) dw[wi] += g * scale;
                if (gamma) acc += g * w[wi];
            }
        }
    } else if (RS == 1 && (Cin & 3) == 0 && (CinP & 3) == 0) {
        for (int j = threadIdx.x * 4; j < per; j += blockDim.x * 4) {     // 1x1 / Linear: same order on both sides
            const float4 g = ld4s(dwp + (long)n * CinP + j);
            const long wi = (long)n * per + j;
            if (dw) {
                float4 o = *reinterpret_cast<float4*>(dw + wi);
                o.x += g.x * scale; o.y += g.y * scale; o.z += g.z * scale; o.w += g.w * scale;
                *reinterpret_cast<float4*>(dw + wi) = o;
            }
            if (gamma) {
                const float4 ww = *reinterpret_cast<const float4*>(w + wi);
                acc += g.x * ww.x + g.y * ww.y + g.z * ww.z + g.w * ww.w;
            }
        }
    } else
    for (int j = threadIdx.x; j < per; j += blockDim.x) {
        const int c = j / RS, t = j - c * RS;       // j indexes dw[n][c][t] (coalesced writes)
        const float g = ld1(dwp + ((long)t * CoutP + n) * CinP + c);
        const long wi = (long)n * per + j;
        if (dw) dw[wi] += g * scale;
        if (gamma) acc += g * w[wi];
    }
    if (gamma) {
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float s = part[0] + part[1] + part[2] + part[3];
            const float dbn = db ? db[n] : 0.f;
            if (dgamma) dgamma[n] += s * rs - dbn * mean[n] * rs;
            if (dbeta) dbeta[n] += dbn;
        }
    } else if (threadIdx.x == 0 && dbeta && db) {
        dbeta[n] += db[n];
    }
}
LOFT_EXPORT int loft_fold_unpack_bwd_multi(const int64_t* desc, int njobs, int64_t nblocks, int lds_floats, void* stream) {
    if (njobs <= 0 || nblocks <= 0) return 0;
    if (lds_floats < 0 || lds_floats > 16384) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_unpack_bwd_multi_kernel, dim3((unsigned)nblocks), dim3(256), (size_t)lds_floats * 4, (hipStream_t)stream,
                       (const long*)desc, njobs, lds_floats);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_fold_unpack_bwd(const float* dwp, const float* db, const float* w, const float* gamma, const float* mean,
                                     const float* var, float eps, int Cout, int Cin, int RS, float* dw, float* dgamma,
                                     float* dbeta, int CoutP, int CinP, int accumulate, void* stream) {
    if (Cout <= 0) return 0;
    if (CoutP < Cout || CinP < Cin) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_unpack_bwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dwp, db, w, gamma, mean, var, eps,
                       Cout, Cin, RS, dw, dgamma, dbeta, CoutP, CinP, accumulate);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- backward of a 1x1 head with a handful of outputs
// Replaces, for Cout <= 8 (mask logits: 1, fc_cls + fc_reg: 6, fc_offset: 2), the padded-to-128-channels MFMA route
// (zero fill + strided copy of g, a 128-wide data-gradient GEMM and a weight-gradient GEMM: ~0.6 ms for the 1-channel mask
// logits at 875 x 28 x 28 pixels) by ONE pass over the input:
//     gx[m][c] = (x[m][c] > 0 | no mask) * sum_n g[m][n] * w[n][c]         (bf16, 8 bytes per lane)
//     dw[n][c] += sum_m g[m][n] * x[m][c],   db[n] += sum_m g[m][n]         (fp32 registers -> LDS -> one atomic per entry and block)
// HBM-bound: reads x once (it is also the ReLU mask of the producer), writes gx once.  g fp32 [M][gs] (gs = Cout rounded up
// to 4), x bf16 [M][Cin], w fp32 [Cout][Cin]; Cin % 4 == 0, Cin <= 1024.
template <typename T, int V> struct NhbIO;
template <> struct NhbIO<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) { ld4(p, v); }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) { st4(p, v); }
};
template <> struct NhbIO<bf16_t, 8> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) { ld8(p, v); }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) { st8(p, v); }
};
// fp32 activations (the fp32 parity mode, round 6: the narrow heads took the padded 128-channel GEMM route there -- a zero fill, a
// strided copy, two plane splits of an 822 MB map and two contractions for ONE mask-logit channel): 16-byte accesses = 4 channels
template <> struct NhbIO<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ void st(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

// V channels per lane: 8 (16-byte accesses; the mask-logits launch ran at 2.4 TB/s with 8-byte ones) while the weight and
// accumulator registers allow it (NOUT <= 4), else 4.
template <int NOUT, int V, typename XT = bf16_t>
__global__ __launch_bounds__(256) void narrow_head_bwd_kernel(const float* __restrict__ g, int gs, const XT* __restrict__ x,
                                                              const float* __restrict__ w, long M, int Cin, int relu_in,
                                                              XT* __restrict__ gx, float* __restrict__ dw,
                                                              float* __restrict__ db) {
    const int cg = Cin / V;                        // column groups of V channels
    const int ppi = 256 / cg > 0 ? 256 / cg : 1;   // pixels per block round (Cin = 256, V = 8 -> 8)
    const int tid = threadIdx.x;
    const int col = tid % cg, sub = tid / cg;
    const bool act = sub < ppi;
    const int c0 = col * V;
    float wr[NOUT][V], acc[NOUT][V], bacc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
#pragma unroll
        for (int q = 0; q < V; ++q) { wr[n][q] = w[(long)n * Cin + c0 + q]; acc[n][q] = 0.f; }
        bacc[n] = 0.f;
    }
    if (act) {
        constexpr int UN = 4;                          // pixels in flight per lane: the loop is latency-, not bandwidth-bound
        // each block streams ONE contiguous pixel range (UN * ppi consecutive pixels per round)
        const long step = ppi;
        const long per_blk = ((M + gridDim.x - 1) / gridDim.x + (long)ppi * UN - 1) / ((long)ppi * UN) * ((long)ppi * UN);
        const long m_beg = (long)blockIdx.x * per_blk, m_end = m_beg + per_blk < M ? m_beg + per_blk : M;
        for (long m0 = m_beg + sub; m0 < m_end; m0 += step * UN) {
            float xv[UN][V], gv[UN][NOUT];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long m = m0 + u * step;
                if (m < m_end) {
                    NhbIO<XT, V>::ld(x + m * Cin + c0, xv[u]);
#pragma unroll
                    for (int n = 0; n < NOUT; ++n) gv[u][n] = g[m * gs + n];
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) xv[u][q] = 0.f;
#pragma unroll
                    for (int n = 0; n < NOUT; ++n) gv[u][n] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long m = m0 + u * step;
                float o[V];
#pragma unroll
                for (int q = 0; q < V; ++q) o[q] = 0.f;
#pragma unroll
                for (int n = 0; n < NOUT; ++n) {
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        o[q] += gv[u][n] * wr[n][q];
                        acc[n][q] += gv[u][n] * xv[u][q];
                    }
                    bacc[n] += gv[u][n];
                }
                if (gx && m < m_end) {
                    if (relu_in) {
#pragma unroll
                        for (int q = 0; q < V; ++q) o[q] = xv[u][q] > 0.f ? o[q] : 0.f;
                    }
                    NhbIO<XT, V>::st(gx + m * Cin + c0, o);
                }
            }
        }
    }
    // block reduction over the ppi pixel sub-streams, then one atomic per (n, c) and block
    __shared__ float red[NOUT][1024];
    for (int s2 = 0; s2 < ppi; ++s2) {
        if (act && sub == s2) {
#pragma unroll
            for (int n = 0; n < NOUT; ++n)
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    if (s2 == 0) red[n][c0 + q] = acc[n][q];
                    else red[n][c0 + q] += acc[n][q];
                }
        }
        __syncthreads();
    }
    if (dw)
        for (int i = tid; i < NOUT * Cin; i += 256) unsafeAtomicAdd(dw + i, red[i / Cin][i % Cin]);
    if (db && col == 0 && act) {
#pragma unroll
        for (int n = 0; n < NOUT; ++n) unsafeAtomicAdd(db + n, bacc[n]);
    }
}
LOFT_EXPORT int loft_narrow_head_bwd(const float* g, int g_stride, const void* x, const float* w, int64_t M, int Cin, int Cout,
                                     int relu_in, void* gx, float* dw, float* db, void* stream) {
    if (M <= 0) return 0;
    if (Cout < 1 || Cout > 8 || (Cin & 3) || Cin > 1024 || g_stride < Cout) return (int)hipErrorInvalidValue;
    const int V = (Cout <= 4 && (Cin & 7) == 0) ? 8 : 4;
    const int cg = Cin / V, ppi = 256 / cg > 0 ? 256 / cg : 1;
    // >= 16 pixel rounds per block (each block ends with an LDS reduction and Cout * Cin atomics), at most 2048 blocks
    // (measured: 8192 blocks +0.7 ms per step, 512..2048 equal)
    long blocks = (M + (long)ppi * 16 - 1) / ((long)ppi * 16);
    const long cap = 2048;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
#define NHB(N, VV) hipLaunchKernelGGL((narrow_head_bwd_kernel<N, VV>), dim3((unsigned)blocks), dim3(256), 0, s, g, g_stride, \
                                      (const bf16_t*)x, w, (long)M, Cin, relu_in, (bf16_t*)gx, dw, db)
    if (V == 8) {
        switch (Cout) { case 1: NHB(1, 8); break; case 2: NHB(2, 8); break; case 3: NHB(3, 8); break; default: NHB(4, 8); break; }
    } else {
        switch (Cout) {
            case 1: NHB(1, 4); break; case 2: NHB(2, 4); break; case 3: NHB(3, 4); break; case 4: NHB(4, 4); break;
            case 5: NHB(5, 4); break; case 6: NHB(6, 4); break; case 7: NHB(7, 4); break; default: NHB(8, 4); break;
        }
    }
#undef NHB
    LOFT_LAUNCH_CHECK();
    return 0;
}

// the same pass on fp32 activations (x, gx fp32 [M][Cin]): the fp32 parity mode's narrow heads
LOFT_EXPORT int loft_narrow_head_bwd_f32(const float* g, int g_stride, const float* x, const float* w, int64_t M, int Cin, int Cout,
                                         int relu_in, float* gx, float* dw, float* db, void* stream) {
    if (M <= 0) return 0;
    if (Cout < 1 || Cout > 8 || (Cin & 3) || Cin > 1024 || g_stride < Cout) return (int)hipErrorInvalidValue;
    const int cg = Cin / 4, ppi = 256 / cg > 0 ? 256 / cg : 1;
    long blocks = (M + (long)ppi * 16 - 1) / ((long)ppi * 16);
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
#define NHBF(N) hipLaunchKernelGGL((narrow_head_bwd_kernel<N, 4, float>), dim3((unsigned)blocks), dim3(256), 0, s, g, g_stride, x, w, \
                                   (long)M, Cin, relu_in, gx, dw, db)
    switch (Cout) {
        case 1: NHBF(1); break; case 2: NHBF(2); break; case 3: NHBF(3); break; case 4: NHBF(4); break;
        case 5: NHBF(5); break; case 6: NHBF(6); break; case 7: NHBF(7); break; default: NHBF(8); break;
    }
#undef NHBF
    LOFT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- losses: value and gradient in one launch
// mmdet/models/losses: weight_reduce_loss (utils.py:26-52) around L1 / SmoothL1 (smooth_l1_loss.py:8-50), sigmoid and
// softmax cross-entropy (cross_entropy_loss.py:9-125).  loss = scale / denom * sum_i w_i * l(pred_i, target_i) and
// grad_i = scale / denom * w_i * dl/dpred_i, with denom = *avg_factor (device scalar) or `count` -- replaces the ~10 forward
// and ~10 autograd elementwise launches of every loss by one launch (+ one multiply by the incoming scalar in backward).
// mode 0 L1, 1 SmoothL1(beta), 2 BCE-with-logits (target in [0,1]), 3 softmax CE (pred [n,C], target = int64 class per row,
// weight per row).  Block partial sums go to `partial`; the last block (ticket in `counter`, reset for the next launch) adds
// them in a fixed order: deterministic, no pre-zeroed output.
// v2 additions (one launch where the host side needed a clone / cast / compare launch per operand before):
//   * pred may be a strided VIEW of a wider head output (the [n, 2] class columns of the fused [n, 8] fc_cls + fc_reg output, one
//     channel of a 4-padded NHWC map, rows b, p < P of a [B, S, 5] gather): logical index i (element; row for mode 3) sits at
//     (i / (d1 d2)) s0 + ((i / d2) % d1) s1 + (i % d2) s2 floats; mode 3 reads its C classes at unit stride from there;
//   * target_kind 1: int64 labels, BCE target = (label >= 1) (the RPN's {0, 1} labels, cross_entropy_loss.py:60-66);
//   * weight_kind 1: uint8 / bool weights; every `wdiv` consecutive logical elements share one weight (a per-row weight
//     broadcast over the 4 box deltas);
//   * mode 3 with want_acc: loss_out[1] = top-1 accuracy in percent (accuracy.py:4-48: 100 * #(argmax == label) / n, first
//     maximum on ties like torch.argmax), counted in partial[256 + block].
struct LossView { long d1, d2, s0, s1, s2; };
__device__ __forceinline__ long loss_view_off(const LossView& v, long i) {
    if (v.d1 == 1 && v.d2 == 1) return i * v.s0;
    const long i2 = i % v.d2, r = i / v.d2;
    return (r / v.d1) * v.s0 + (r % v.d1) * v.s1 + i2 * v.s2;
}
__global__ __launch_bounds__(256) void fused_loss_kernel(int mode, const float* __restrict__ pred, LossView pv, const void* __restrict__ target,
                                                         int target_kind, const void* __restrict__ weight, int weight_kind, long wdiv,
                                                         long n, int C, const float* avg_factor,
                                                         float count, float scale, float beta, float* __restrict__ grad,
                                                         float* __restrict__ partial, unsigned* __restrict__ counter,
                                                         float* __restrict__ loss_out, int want_acc) {
    const float denom = avg_factor ? *avg_factor : count;
    const float k = scale / denom;
    float acc = 0.f, hit = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float w = 1.f;
        if (weight) {
            const long wi = wdiv > 1 ? i / wdiv : i;
            w = weight_kind ? (reinterpret_cast<const unsigned char*>(weight)[wi] ? 1.f : 0.f) : reinterpret_cast<const float*>(weight)[wi];
        }
        if (mode == 3) {
            const float* p = pred + loss_view_off(pv, i);
            const long lab = reinterpret_cast<const int64_t*>(target)[i];
            float mx = p[0];
            int am = 0;
            for (int c = 1; c < C; ++c) if (p[c] > mx) { mx = p[c]; am = c; }
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
            const float lse = mx + logf(se);
            acc += w * (lse - p[lab]);
            hit += am == lab ? 1.f : 0.f;
            for (int c = 0; c < C; ++c) grad[i * C + c] = k * w * (expf(p[c] - lse) - (c == lab ? 1.f : 0.f));
        } else {
            const float p = pred[loss_view_off(pv, i)];
            const float t = target_kind == 1 ? (reinterpret_cast<const int64_t*>(target)[i] >= 1 ? 1.f : 0.f)
                                             : reinterpret_cast<const float*>(target)[i];
            float l, g;
            if (mode == 2) {
                // max(p,0) - p t + log(1 + exp(-|p|))   (torch's binary_cross_entropy_with_logits form)
                l = fmaxf(p, 0.f) - p * t + log1pf(expf(-fabsf(p)));
                g = 1.f / (1.f + expf(-p)) - t;
            } else {
                const float d = p - t, ad = fabsf(d);
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                if (mode == 0) { l = ad; g = sg; }
                else if (ad < beta) { l = 0.5f * ad * ad / beta; g = d / beta; }
                else { l = ad - 0.5f * beta; g = sg; }
            }
            acc += w * l;
            grad[i] = k * w * g;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_down(acc, o, 64); hit += __shfl_down(hit, o, 64); }
    __shared__ float part[4], hpart[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; hpart[threadIdx.x >> 6] = hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        if (want_acc) partial[256 + blockIdx.x] = hpart[0] + hpart[1] + hpart[2] + hpart[3];
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float tot = 0.f, hits = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) tot += reinterpret_cast<volatile float*>(partial)[b];
        *loss_out = tot * k;
        if (want_acc) {
            for (unsigned b = 0; b < gridDim.x; ++b) hits += reinterpret_cast<volatile float*>(partial)[256 + b];
            loss_out[1] = n > 0 ? hits * (100.0f / (float)n) : 0.f;
        }
        *counter = 0u;
    }
}
LOFT_EXPORT int loft_fused_loss_v2(int mode, const float* pred, int64_t d1, int64_t d2, int64_t s0, int64_t s1, int64_t s2,
                                   const void* target, int target_kind, const void* weight, int weight_kind, int64_t wdiv, int64_t n, int C,
                                   const float* avg_factor, float count, float scale, float beta, float* grad, float* partial,
                                   uint32_t* counter, float* loss_out, int want_acc, void* stream) {
    if (mode < 0 || mode > 3 || n < 0 || (mode == 3 && C < 1) || !partial || !counter || d1 < 1 || d2 < 1 || wdiv < 1 ||
        target_kind < 0 || target_kind > 1 || (target_kind == 1 && mode != 2) || weight_kind < 0 || weight_kind > 1 ||
        (want_acc && mode != 3))
        return (int)hipErrorInvalidValue;
    long blocks = n <= 0 ? 1 : (n + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    LossView pv{(long)d1, (long)d2, (long)s0, (long)s1, (long)s2};
    hipLaunchKernelGGL(fused_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mode, pred, pv, target, target_kind,
                       weight, weight_kind, (long)wdiv, (long)n, C, avg_factor, count, scale, beta, grad, partial, counter, loss_out,
                       want_acc);
    LOFT_LAUNCH_CHECK();
    return 0;
}
LOFT_EXPORT int loft_fused_loss(int mode, const float* pred, const void* target, const float* weight, int64_t n, int C,
                                const float* avg_factor, float count, float scale, float beta, float* grad, float* partial,
                                uint32_t* counter, float* loss_out, void* stream) {
    return loft_fused_loss_v2(mode, pred, 1, 1, mode == 3 ? C : 1, 0, 0, target, 0, weight, 0, 1, n, C, avg_factor, count, scale, beta,
                              grad, partial, counter, loss_out, 0, stream);
}
