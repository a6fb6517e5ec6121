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

LOFT_EXPORT int loft_conv_wgrad_f32(const float* g, const float* x, float* dw, int B, int GH, int GW, int Cout, int XH, int XW, int Cin,
                                    int OH, int OW, int gos, int ss, int T, const int* goy_host, const int* gox_host,
                                    const int* dy_host, const int* dx_host, const int* wt_host, int groups, int64_t g_gs,
                                    int64_t x_gs, int64_t dw_gs, void* stream) {
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
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    long splits = 2048 / ((long)tiles * T * groups);
    const long maxs = (M + 255) / 256;
    splits = splits < 1 ? 1 : (splits > maxs ? maxs : splits);
    int pps = (int)((M + splits - 1) / splits);
    pps = (pps + 31) / 32 * 32;
    a.pix_per_split = pps;
    dim3 grid(tiles, T * groups, (unsigned)((M + pps - 1) / pps));
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

__global__ void relu_bwd_f32_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i], yv = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(yv.x > 0.f ? gv.x : 0.f, yv.y > 0.f ? gv.y : 0.f, yv.z > 0.f ? gv.z : 0.f,
                                                       yv.w > 0.f ? gv.w : 0.f);
    }
}
LOFT_EXPORT int loft_relu_bwd_f32(const float* g, const float* y, float* out, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 4) return (int)hipErrorInvalidValue;
    const long n4 = n / 4;
    hipLaunchKernelGGL(relu_bwd_f32_kernel, dim3((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096)), dim3(256), 0,
                       (hipStream_t)stream, g, y, out, n4);
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
