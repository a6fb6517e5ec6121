// parity_f32.hip -- fp32 forms of the BACKWARD pieces of the tap-conv path, for the fp32 parity mode (forward: loft_conv_tap_f32,
// conv_mfma.hip).  They exist so that parameter gradients of the whole model can be compared with the reference's fp32 autograd at
// the north-star tolerance (1e-3); they are a checker path, not a performance path.
//   loft_conv_wgrad_f32     weight gradient of the tap convolution, contraction on v_mfma_f32_32x32x2_f32 (exact fp32 products/sums)
//   loft_relu_bwd_f32       out = g * (y > 0)
//   loft_downsum2x_add_f32  coarse += 2x2 block sums of fine          (adjoint of the FPN top-down `+= interpolate(nearest)`, fpn.py:176-188)
//   loft_subsample2_add_f32 big[::2, ::2] += small                    (adjoint of P6 = max_pool2d(P5, 1, stride=2), fpn.py:189-191)
// Reference call sites whose autograd these stand in for: the nn.Conv2d / nn.Linear layers listed in conv_mfma.hip.
#include "conv_tap.h"
#include "../../include/loft_hip.h"

typedef __attribute__((ext_vector_type(16))) float pf32x16;

struct WgradF32Args {
    const float* g; const float* x; float* dw;
    int B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T;
    int goy[CONV_MAX_TAPS], gox[CONV_MAX_TAPS], dy[CONV_MAX_TAPS], dx[CONV_MAX_TAPS], wt[CONV_MAX_TAPS];
    long g_gs, x_gs, dw_gs;
    int M, pix_per_split;
};

// dW[wt[t]][n][c] += sum_m G[b, oy*gos+goy, ox*gos+gox, n] * X[b, oy*ss+dy, ox*ss+dx, c].  Workgroup = 64 (n) x 64 (c) tile of one
// tap, one pixel range; 4 waves of one 32x32 accumulator; 32-pixel K-steps staged through LDS with plain loads.
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const WgradF32Args a) {
    __shared__ float gs[32][64 + 1], xs[32][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctiles = (a.Cin + 63) / 64;
    const int nt = blockIdx.x / ctiles, ct = blockIdx.x - nt * ctiles;
    const int t = blockIdx.y % a.T, grp = blockIdx.y / a.T;
    const int mbeg = blockIdx.z * a.pix_per_split, mend = min(a.M, mbeg + a.pix_per_split);
    if (mbeg >= mend) return;
    const float* G = a.g + (long)grp * a.g_gs;
    const float* X = a.x + (long)grp * a.x_gs;
    const int ohw = a.OH * a.OW;
    const int n0 = nt * 64, c0 = ct * 64;
    const int wn = wave >> 1, wc = wave & 1;
    pf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int m0 = mbeg; m0 < mend; m0 += 32) {
        __syncthreads();
        // stage 32 pixels x 64 channels of each operand: thread -> pixel tid/8, channels (tid%8)*8 .. +8
        {
            const int pr = tid >> 3, cc = (tid & 7) * 8;
            const int m = m0 + pr;
            bool ok = m < mend;
            long gpix = 0, xpix = 0;
            if (ok) {
                const int b = m / ohw, rem = m - b * ohw;
                const int oy = rem / a.OW, ox = rem - oy * a.OW;
                const int gy = oy * a.gos + a.goy[t], gx = ox * a.gos + a.gox[t];
                const int iy = oy * a.ss + a.dy[t], ix = ox * a.ss + a.dx[t];
                ok = (gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW);
                gpix = ((long)b * a.GH + gy) * a.GW + gx;
                xpix = ((long)b * a.XH + iy) * a.XW + ix;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = n0 + cc + e, c = c0 + cc + e;
                gs[pr][cc + e] = (ok && n < a.Cout) ? G[gpix * a.Cout + n] : 0.f;
                xs[pr][cc + e] = (ok && c < a.Cin) ? X[xpix * a.Cin + c] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            const float av = gs[k + (lane >> 5)][wn * 32 + (lane & 31)];
            const float bv = xs[k + (lane >> 5)][wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    float* dw = a.dw + (long)grp * a.dw_gs + (long)a.wt[t] * a.Cout * a.Cin;
    const int c = c0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < a.Cout && c < a.Cin) unsafeAtomicAdd(dw + (long)n * a.Cin + c, acc[r]);
    }
}

// ---- the same weight gradient with split-bf16 operands (LOFT_F32_SPLIT3, the mode's default since round 4) --------------------
// g = gh + gl, x = xh + xl (two bf16 each: 16 mantissa bits, see conv_mfma.hip f32_split2); dW += gh xh + gh xl + gl xh on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation.  Workgroup = 128 (n) x 128 (c) tile of one tap and one range of reduction rows,
// 4 waves of 64 x 64; a K-step = 32 reduction rows.  The split happens ONCE per element on the way into LDS (the exact kernel's
// 64 x 64 tile with scalar loads re-read every element eight times and ran at ~44 TFLOP/s): a thread loads 16 consecutive
// channels of one row of G and of X (the next K-step's rows are requested before the current one is consumed), writes their hi
// and lo halves as 16-byte chunks of four bf16 tiles [32 rows][128 ch] (256-byte rows, chunk q of row r at q ^ ((r & 3) << 2):
// the layout of the transposing LDS reader ds_read_tr16_b64, "8 consecutive rows for my column", as in roi_align.hip).
typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 pbf16x2;
typedef __attribute__((ext_vector_type(4))) short ps16x4;
__device__ __forceinline__ void pf32_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const loft_f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pbf16x2));
    const loft_f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, pbf16x2));
}
__device__ __forceinline__ int pswz(int row, int q) { return q ^ ((row & 3) << 2); }
// lane l -> column col0 + (l & 31), rows kbase + 8 * (l >> 5) + 0..7 of a row-major [k][128] bf16 tile (256-byte rows)
__device__ __forceinline__ pbf16x8 ptr_frag(const char* tile, int kbase, int col0, int lane) {
    const int il = lane & 15, gl = lane >> 4;
    const int col = col0 + 16 * (gl & 1) + (il & 3) * 4;
    const int r0 = kbase + 8 * (gl >> 1) + (il >> 2), r1 = r0 + 4;
    const char* p0 = tile + r0 * 256 + pswz(r0, col >> 3) * 16 + (col & 7) * 2;
    const char* p1 = tile + r1 * 256 + pswz(r1, col >> 3) * 16 + (col & 7) * 2;
    const ps16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)p0);
    const ps16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)p1);
    typedef __attribute__((ext_vector_type(8))) short ps16x8;
    const ps16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(pbf16x8, f);
}

template <int NS>         // bf16 pieces per fp32 operand: 2 (three product terms) or 3 (six)
__global__ __launch_bounds__(256) void conv_wgrad_f32x3_kernel(const WgradF32Args a) {
    constexpr int TB = 32 * 256;                                  // one bf16 tile [32][128]
    __shared__ __attribute__((aligned(16))) char lds[2 * NS * TB];     // G hi, (mid,) lo, X hi, (mid,) lo
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctiles = (a.Cin + 127) / 128;
    const int nt = blockIdx.x / ctiles, ct = blockIdx.x - nt * ctiles;
    const int t = blockIdx.y % a.T, grp = blockIdx.y / a.T;
    const int mbeg = blockIdx.z * a.pix_per_split, mend = min(a.M, mbeg + a.pix_per_split);
    if (mbeg >= mend) return;
    const float* G = a.g + (long)grp * a.g_gs;
    const float* X = a.x + (long)grp * a.x_gs;
    const int ohw = a.OH * a.OW;
    const int n0 = nt * 128, c0 = ct * 128;
    const int wn = wave >> 1, wc = wave & 1;
    const int pr = tid >> 3, cc = (tid & 7) * 16;                 // staging: row pr of the K-step, channels cc .. cc + 15 of the tile
    const int goy = a.goy[t], gox = a.gox[t], dy = a.dy[t], dx = a.dx[t];
    const bool gvec = !(a.Cout & 3), xvec = !(a.Cin & 3);
    pf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float gr[16], xr[16];
    auto fetch = [&](int m0) {                                    // rows of the K-step starting at m0 -> registers
        const int m = m0 + pr;
        bool ok = m < mend;
        long gpix = 0, xpix = 0;
        if (ok) {
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            const int gy = oy * a.gos + goy, gx = ox * a.gos + gox;
            const int iy = oy * a.ss + dy, ix = ox * a.ss + dx;
            ok = (gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW);
            gpix = ((long)b * a.GH + gy) * a.GW + gx;
            xpix = ((long)b * a.XH + iy) * a.XW + ix;
        }
        const float* gp = G + gpix * a.Cout + n0 + cc;
        const float* xp = X + xpix * a.Cin + c0 + cc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = n0 + cc + 4 * v, c = c0 + cc + 4 * v;
            if (ok && gvec && n + 3 < a.Cout) {
                const float4 q = *reinterpret_cast<const float4*>(gp + 4 * v);
                gr[4 * v] = q.x; gr[4 * v + 1] = q.y; gr[4 * v + 2] = q.z; gr[4 * v + 3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) gr[4 * v + e] = (ok && n + e < a.Cout) ? gp[4 * v + e] : 0.f;
            }
            if (ok && xvec && c + 3 < a.Cin) {
                const float4 q = *reinterpret_cast<const float4*>(xp + 4 * v);
                xr[4 * v] = q.x; xr[4 * v + 1] = q.y; xr[4 * v + 2] = q.z; xr[4 * v + 3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[4 * v + e] = (ok && c + e < a.Cin) ? xp[4 * v + e] : 0.f;
            }
        }
    };
    auto put = [&](const float (&v)[16], char* base) {             // 16 channels -> two 16-byte chunks of each of the NS tiles
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t pc[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r0 = v[8 * h + 2 * e], r1 = v[8 * h + 2 * e + 1];
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) {
                    const loft_f32x2 rv = {r0, r1};
                    pc[s_][e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rv, pbf16x2));
                    r0 -= __uint_as_float(pc[s_][e] << 16); r1 -= __uint_as_float(pc[s_][e] & 0xffff0000u);
                }
            }
            const int off = pr * 256 + pswz(pr, (cc >> 3) + h) * 16;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) *reinterpret_cast<uint4*>(base + s_ * TB + off) = make_uint4(pc[s_][0], pc[s_][1], pc[s_][2], pc[s_][3]);
        }
    };
    fetch(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += 32) {
        __syncthreads();                                          // the previous K-step's fragment reads are done
        put(gr, lds);
        put(xr, lds + NS * TB);
        __syncthreads();
        if (m0 + 32 < mend) fetch(m0 + 32);                       // in flight during this K-step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pbf16x8 gp[NS][2], xp[NS][2];                         // [piece: hi, (mid,) lo][fragment]
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    gp[s_][i] = ptr_frag(lds + s_ * TB, ks * 16, wn * 64 + i * 32, lane);
                    xp[s_][i] = ptr_frag(lds + (NS + s_) * TB, ks * 16, wc * 64 + i * 32, lane);
                }
            // product terms whose pieces' indices sum to <= NS - 1, smallest first (NS = 2: hl, lh, hh; NS = 3: hl, lh, mm, hm, mh, hh)
#pragma unroll
            for (int lvl = NS - 1; lvl >= 0; --lvl)
#pragma unroll
                for (int sa = 0; sa <= lvl; ++sa)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp[sa][i], xp[lvl - sa][j], acc[i][j], 0, 0, 0);
        }
    }
    float* dw = a.dw + (long)grp * a.dw_gs + (long)a.wt[t] * a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < a.Cout && c < a.Cin) unsafeAtomicAdd(dw + (long)n * a.Cin + c, acc[i][j][r]);
            }
        }
}

LOFT_EXPORT int loft_conv_wgrad_f32_v(const float* g, const float* x, float* dw, int B, int GH, int GW, int Cout, int XH, int XW, int Cin,
                                      int OH, int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host,
                                      const int* dy_host, const int* dx_host, const int* wt_host, int groups, int64_t g_gs,
                                      int64_t x_gs, int64_t dw_gs, int variant, void* stream) {
    if (variant != LOFT_F32_SPLIT6 && variant != LOFT_F32_SPLIT3 && variant != LOFT_F32_EXACT) return (int)hipErrorInvalidValue;
    if (T < 1 || T > CONV_MAX_TAPS || groups < 1) return (int)hipErrorInvalidValue;
    WgradF32Args a;
    a.g = g; a.x = x; a.dw = dw;
    a.B = B; a.GH = GH; a.GW = GW; a.Cout = Cout; a.XH = XH; a.XW = XW; a.Cin = Cin; a.OH = OH; a.OW = OW; a.gos = gos; a.ss = ss; a.T = T;
    for (int t = 0; t < T; ++t) {
        a.goy[t] = goy_host[t]; a.gox[t] = gox_host[t]; a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t];
    }
    a.g_gs = g_gs; a.x_gs = x_gs; a.dw_gs = dw_gs;
    const long M = (long)B * OH * OW;
    if (M <= 0) return 0;
    if (M > 0x7fffffffL) return (int)hipErrorInvalidValue;
    a.M = (int)M;
    const int tw = variant == LOFT_F32_EXACT ? 64 : 128;
    const int tiles = ((Cout + tw - 1) / tw) * ((Cin + tw - 1) / tw);
    long splits = 2048 / ((long)tiles * T * groups);
    const long maxs = (M + 255) / 256;
    splits = splits < 1 ? 1 : (splits > maxs ? maxs : splits);
    int pps = (int)((M + splits - 1) / splits);
    pps = (pps + 31) / 32 * 32;
    a.pix_per_split = pps;
    dim3 grid(tiles, T * groups, (unsigned)((M + pps - 1) / pps));
    if (variant == LOFT_F32_EXACT) hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (variant == LOFT_F32_SPLIT3) hipLaunchKernelGGL(conv_wgrad_f32x3_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_wgrad_f32x3_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_conv_wgrad_f32(const float* g, const float* x, float* dw, int B, int GH, int GW, int Cout, int XH, int XW, int Cin,
                                    int OH, int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host,
                                    const int* dy_host, const int* dx_host, const int* wt_host, int groups, int64_t g_gs,
                                    int64_t x_gs, int64_t dw_gs, void* stream) {
    return loft_conv_wgrad_f32_v(g, x, dw, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host, gox_host, dy_host, dx_host,
                                 wt_host, groups, g_gs, x_gs, dw_gs, LOFT_F32_SPLIT6, stream);
}

// amax (may be NULL; PRE-ZEROED): max |out| as a device scalar -- the masked gradient goes straight into a plane split, which then
// skips its absmax pass (round 6)
__global__ __launch_bounds__(256) void relu_bwd_f32_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out,
                                                           long n4, unsigned* __restrict__ amax) {
    float m = 0.f;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i], yv = reinterpret_cast<const float4*>(y)[i];
        const float4 o = make_float4(yv.x > 0.f ? gv.x : 0.f, yv.y > 0.f ? gv.y : 0.f, yv.z > 0.f ? gv.z : 0.f, yv.w > 0.f ? gv.w : 0.f);
        reinterpret_cast<float4*>(out)[i] = o;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        bad |= (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (o.w != o.w);
    }
    if (amax) {
        if (bad) m = __builtin_inff();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        __shared__ float wm[4];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            if (m > 0.f) atomicMax(amax, __float_as_uint(m));
        }
    }
}
LOFT_EXPORT int loft_relu_bwd_f32(const float* g, const float* y, float* out, int64_t n, float* amax_out, void* stream) {
    if (n <= 0) return 0;
    if (n % 4) return (int)hipErrorInvalidValue;
    const long n4 = n / 4;
    hipLaunchKernelGGL(relu_bwd_f32_kernel, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, g, y, out, n4, (unsigned*)amax_out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// coarse[b, y, x, c] += fine[b, 2y, 2x, c] + fine[b, 2y, 2x+1, c] + fine[b, 2y+1, 2x, c] + fine[b, 2y+1, 2x+1, c] (odd edges clipped)
__global__ void downsum2x_add_f32_kernel(float* __restrict__ coarse, const float* __restrict__ fine, int B, int Hc, int Wc, int Hf,
                                         int Wf, int C) {
    const long n = (long)B * Hc * Wc * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long p = i / C;
        const int x = (int)(p % Wc); p /= Wc;
        const int y = (int)(p % Hc);
        const int b = (int)(p / Hc);
        float s = 0.f;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int fy = 2 * y + dy, fx = 2 * x + dx;
                if (fy < Hf && fx < Wf) s += fine[(((long)b * Hf + fy) * Wf + fx) * C + c];
            }
        coarse[i] += s;
    }
}
LOFT_EXPORT int loft_downsum2x_add_f32(float* coarse, const float* fine, int B, int Hc, int Wc, int Hf, int Wf, int C, void* stream) {
    const long n = (long)B * Hc * Wc * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(downsum2x_add_f32_kernel, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, coarse, fine, B, Hc, Wc, Hf, Wf, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}

__global__ void subsample2_add_f32_kernel(float* __restrict__ big, const float* __restrict__ small, int B, int Hs, int Ws, int Hb,
                                          int Wb, int C) {
    const long n = (long)B * Hs * Ws * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long p = i / C;
        const int x = (int)(p % Ws); p /= Ws;
        const int y = (int)(p % Hs);
        const int b = (int)(p / Hs);
        big[(((long)b * Hb + 2 * y) * Wb + 2 * x) * C + c] += small[i];
    }
}
LOFT_EXPORT int loft_subsample2_add_f32(float* big, const float* small, int B, int Hs, int Ws, int Hb, int Wb, int C, void* stream) {
    const long n = (long)B * Hs * Ws * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(subsample2_add_f32_kernel, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, big, small, B, Hs, Ws, Hb, Wb, C);
    LOFT_LAUNCH_CHECK();
    return 0;
}
