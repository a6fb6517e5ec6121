// rpn_sparse.hip -- row gather / scatter for the sparse backward of the RPN head (gfx950).
//
// The RPN losses (mmdet/models/dense_heads/anchor_head.py:429-497 via rpn_head.py:56-80) touch only the sampled anchors --
// at most 256 per image (RandomSampler num=256, cfg train_cfg.rpn.sampler) of 261 888 at 1024^2 -- so the gradient of the
// head's outputs is zero everywhere else.  Instead of running dense data/weight-gradient convolutions over five pyramid
// levels for a gradient that is >99.8 % zeros, the backward works on the selected pixels only:
//   gather   : rows x[level][b, y+dy, x+dx, :] of the K x K neighbourhood of every selected pixel -> [nsel][K*K][C]
//              (the A operand of a dense [nsel x K*K*C] GEMM for the weight gradient)
//   scatter  : dx[level][b, y+dy, x+dx, :] += src[i][tap][:]  (packed bf16 atomics: neighbourhoods of nearby anchors overlap)
// Both are HBM/L2-bound row copies of 512 B (C = 256 bf16); nsel * K*K rows in total (<= 36 864 per step).
#include "loft_common.h"
#include "../../include/loft_hip.h"

namespace {

struct SparseLevels {
    void* ptr[8];
    int H[8], W[8];
    int n;
};

#ifdef LOFT_ACT_F16
typedef __attribute__((ext_vector_type(2))) _Float16 act16x2_t;
#define LOFT_ATOMIC_ADD_PK16(p, v) __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) act16x2_t*)(p), __builtin_bit_cast(act16x2_t, v))
#else
typedef __attribute__((ext_vector_type(2))) __bf16 act16x2_t;
#define LOFT_ATOMIC_ADD_PK16(p, v) __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) act16x2_t*)(p), __builtin_bit_cast(act16x2_t, v))
#endif

// rows: int32 [nsel][4] = (b, level, y, x); level < 0 marks an inactive row (output zeros / no scatter)
__global__ void rpn_gather_rows_kernel(const SparseLevels lv, const int* __restrict__ rows, int nsel, int C, int K,
                                       bf16_t* __restrict__ out) {
    const int cg = C >> 3, T = K * K, half = K >> 1;
    const long total = (long)nsel * T * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        const long r = i / cg;
        const int t = (int)(r % T), s = (int)(r / T);
        const int4 rw = *reinterpret_cast<const int4*>(rows + (long)s * 4);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (rw.y >= 0 && rw.y < lv.n) {
            const int y = rw.z + t / K - half, x = rw.w + t % K - half;
            const int H = lv.H[rw.y], W = lv.W[rw.y];
            if (y >= 0 && y < H && x >= 0 && x < W)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(lv.ptr[rw.y]) +
                                                    (((long)rw.x * H + y) * W + x) * C + c8 * 8);
        }
        *reinterpret_cast<uint4*>(out + i * 8) = v;
    }
}

__global__ void rpn_scatter_add_kernel(const SparseLevels lv, const int* __restrict__ rows, int nsel, int C, int K,
                                       const bf16_t* __restrict__ src) {
    const int cp = C >> 1, T = K * K, half = K >> 1;
    const long total = (long)nsel * T * cp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % cp);
        const long r = i / cp;
        const int t = (int)(r % T), s = (int)(r / T);
        const int4 rw = *reinterpret_cast<const int4*>(rows + (long)s * 4);
        if (rw.y < 0 || rw.y >= lv.n) continue;
        const int y = rw.z + t / K - half, x = rw.w + t % K - half;
        const int H = lv.H[rw.y], W = lv.W[rw.y];
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const uint32_t v = *reinterpret_cast<const uint32_t*>(src + i * 2);
        if (v == 0u) continue;
        bf16_t* dst = reinterpret_cast<bf16_t*>(lv.ptr[rw.y]) + (((long)rw.x * H + y) * W + x) * C + c2 * 2;
        LOFT_ATOMIC_ADD_PK16(dst, v);
    }
}

// The three operands the sparse backward builds before its GEMMs, in ONE launch (they were ~15 stock launches: cat / arange / zeros /
// scatter_ / casts / permuted copies).  Work items, in this order:
//   g_rows [nsel][P]  : row i = zeros except column slot[i] (= g[i][0]) and columns A + 4 slot[i] + j (= g[i][1 + j]) -- the output
//                       gradient in the fused head's channel order (objectness 0..A-1, deltas A..5A-1), padded to P channels
//   w_headT [C][P]    : w_headT[c][n] = w_cls[n][c] (n < A), w_reg[n - A][c] (A <= n < 5A), 0 beyond: the head's dgrad operand
//   wd [9 C][C]       : wd[t C + ci][co] = w_conv[co][ci][t]: the 3x3 conv's dgrad operand, (tap, cin)-major
__global__ void rpn_sparse_prep_kernel(const float* __restrict__ g, const long* __restrict__ slot, int nsel, int A, int P, int C,
                                       const float* __restrict__ w_cls, const float* __restrict__ w_reg,
                                       const float* __restrict__ w_conv, bf16_t* __restrict__ g_rows,
                                       bf16_t* __restrict__ w_headT, bf16_t* __restrict__ wd) {
    const long n0 = (long)nsel * P, n1 = n0 + (long)C * P, n2 = n1 + 9L * C * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        if (i < n0) {
            const int r = (int)(i / P), n = (int)(i - (long)r * P);
            const int a = (int)slot[r];
            float v = 0.f;
            if (n == a) v = g[(long)r * 5];
            else if (n >= A + 4 * a && n < A + 4 * a + 4) v = g[(long)r * 5 + 1 + (n - A - 4 * a)];
            g_rows[i] = f32_to_bf16(v);
        } else if (i < n1) {
            const long j = i - n0;
            const int c = (int)(j / P), n = (int)(j - (long)c * P);
            const float v = n < A ? w_cls[(long)n * C + c] : (n < 5 * A ? w_reg[(long)(n - A) * C + c] : 0.f);
            w_headT[j] = f32_to_bf16(v);
        } else {
            const long j = i - n1;
            const int co = (int)(j % C);
            const long r = j / C;
            const int ci = (int)(r % C), t = (int)(r / C);
            wd[j] = f32_to_bf16(w_conv[((long)co * C + ci) * 9 + t]);
        }
    }
}

int fill_levels(SparseLevels* lv, void* const* ptrs, const int* H, const int* W, int n) {
    if (n < 1 || n > 8) return (int)hipErrorInvalidValue;
    lv->n = n;
    for (int i = 0; i < 8; ++i) { lv->ptr[i] = i < n ? ptrs[i] : nullptr; lv->H[i] = i < n ? H[i] : 0; lv->W[i] = i < n ? W[i] : 0; }
    return 0;
}

}  // namespace

LOFT_EXPORT int loft_rpn_gather_rows(void* const* level_ptrs, const int* H, const int* W, int n_levels, const int* rows, int nsel,
                                     int C, int K, void* out, void* stream) {
    SparseLevels lv;
    if (int e = fill_levels(&lv, level_ptrs, H, W, n_levels)) return e;
    if ((C % 8) || !(K & 1)) return (int)hipErrorInvalidValue;
    const long total = (long)nsel * K * K * (C / 8);
    if (total <= 0) return 0;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rpn_gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lv, rows, nsel, C, K,
                       (bf16_t*)out);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_rpn_scatter_add_rows(void* const* level_ptrs, const int* H, const int* W, int n_levels, const int* rows,
                                          int nsel, int C, int K, const void* src, void* stream) {
    SparseLevels lv;
    if (int e = fill_levels(&lv, level_ptrs, H, W, n_levels)) return e;
    if ((C % 2) || !(K & 1)) return (int)hipErrorInvalidValue;
    const long total = (long)nsel * K * K * (C / 2);
    if (total <= 0) return 0;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rpn_scatter_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lv, rows, nsel, C, K,
                       (const bf16_t*)src);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_rpn_sparse_prep(const float* g, const int64_t* slot, int nsel, int A, int P, int C, const float* w_cls,
                                     const float* w_reg, const float* w_conv, void* g_rows, void* w_headT, void* wd, void* stream) {
    if (nsel < 0 || A < 1 || 5 * A > P || C < 1) return (int)hipErrorInvalidValue;
    const long total = (long)nsel * P + (long)C * P + 9L * C * C;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rpn_sparse_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (const long*)slot, nsel, A,
                       P, C, w_cls, w_reg, w_conv, (bf16_t*)g_rows, (bf16_t*)w_headT, (bf16_t*)wd);
    LOFT_LAUNCH_CHECK();
    return 0;
}
