// Probe: what does a 256 x 256 x 64 tile reach with FOUR waves (one per SIMD, wave tile 128 x 128, 256 accumulator registers)
// instead of the eight (two per SIMD, 128 x 64) of conv_pipe.hip?  Plain GEMM  C[m][n] = sum_k X[m][k] * W[n][k]  (both operands
// K-contiguous, like the tap conv's packed weights / NHWC activations), bf16 in, fp32 accumulate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_quad gemm_quad.hip && ./gemm_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <type_traits>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef uint16_t bf16_t;

__device__ __forceinline__ int swz(int row, int q) { return q ^ ((row >> 1) & 7); }
__device__ __forceinline__ int xcd_remap(int L, int N) {
    const int xcd = L & 7, q = N >> 3, r = N & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (L >> 3);
}


__device__ __forceinline__ void zero_acc() {
#define Z(n) asm volatile("v_accvgpr_write_b32 a" #n ", 0" ::: "a" #n);
#define Z8(n) Z(n##0) Z(n##1) Z(n##2) Z(n##3) Z(n##4) Z(n##5) Z(n##6) Z(n##7) Z(n##8) Z(n##9)
    Z(0) Z(1) Z(2) Z(3) Z(4) Z(5) Z(6) Z(7) Z(8) Z(9)
    Z8(1) Z8(2) Z8(3) Z8(4) Z8(5) Z8(6) Z8(7) Z8(8) Z8(9) Z8(10) Z8(11) Z8(12) Z8(13) Z8(14) Z8(15) Z8(16) Z8(17) Z8(18) Z8(19)
    Z8(20) Z8(21) Z8(22) Z8(23) Z8(24)
    Z(250) Z(251) Z(252) Z(253) Z(254) Z(255)
#undef Z8
#undef Z
}

constexpr int WOFF = 0, XOFF = 65536, BUF = 32768, LDS_BYTES = 131072;
#define SB() __builtin_amdgcn_sched_barrier(0)

// LDS: [W b0][W b1][X b0][X b1], tile = 256 rows x 128 B, 16-byte chunk q of row r at position swz(r, q).
template <bool STORE, int ABL>
__global__ __launch_bounds__(256) void gemm_quad_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                       int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = N / 256, ntm = M / 256;
    const int V = xcd_remap(blockIdx.x, ntn * ntm);
    const int m0 = (V / ntn) * 256, n0 = (V % ntn) * 256;
    const int wm = wave >> 1, wn = wave & 1;                 // wave tile: pixels [128 wm, +128), couts [128 wn, +128)
    const int nk = K / 64;

    // staging: a glds instruction = 8 rows x 128 B; wave w issues pieces (i * 4 + w), i = 0..7, of each operand (piece = 8 rows)
    const int lrow = lane >> 3, lchunk = lane & 7;
    const bf16_t* xsrc[8];
    const bf16_t* wsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (i * 4 + wave) * 8 + lrow;
        xsrc[i] = X + (long)(m0 + r) * K + swz(r, lchunk) * 8;
        wsrc[i] = W + (long)(n0 + r) * K + swz(r, lchunk) * 8;
    }
    bool in_loop = false;
    auto issue = [&](int i, int buf, int k0) {          // piece i of both operands of the K-tile at k0
        if ((ABL & 2) && in_loop) return;
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + k0), (lds_ptr_t)(lds + WOFF + buf * BUF + (i * 4 + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + k0), (lds_ptr_t)(lds + XOFF + buf * BUF + (i * 4 + wave) * 1024), 16, 0, 0);
    };

    // fragment addresses: row = base + (lane & 31), chunk = 2 * ks + (lane >> 5)
    const int frow = lane & 31, fq = lane >> 5;
    int woff[4], xoff[4];          // byte offset of (row, chunk 0 + fq) for ks = 0 -- ks adds chunk 2 ks: position (2ks+fq) ^ s(row)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        woff[i] = (wn * 128 + i * 32 + frow) * 128;
        xoff[i] = (wm * 128 + i * 32 + frow) * 128;
    }
    const int sw = (frow >> 1) & 7;                           // swizzle of this lane's rows (row + 32 i has the same (row >> 1) & 7)
    auto rdw = [&](int buf, int ks, int i) {
        if ((ABL & 1) && in_loop) { bf16x8 v; asm volatile("" : "=v"(v)); return v; }
        return *reinterpret_cast<const bf16x8*>(lds + WOFF + buf * BUF + woff[i] + (((2 * ks + fq) ^ sw) << 4));
    };
    auto rdx = [&](int buf, int ks, int i) {
        if ((ABL & 1) && in_loop) { bf16x8 v; asm volatile("" : "=v"(v)); return v; }
        return *reinterpret_cast<const bf16x8*>(lds + XOFF + buf * BUF + xoff[i] + (((2 * ks + fq) ^ sw) << 4));
    };

    // accumulators: acc(i, j) = a[(4 i + j) * 16 .. + 15], touched ONLY by inline asm (hipcc moves a 256-register accumulator
    // set between AGPRs and VGPRs at every loop back-edge when it owns them: 500+ v_accvgpr moves and scratch spills per K-tile)
    zero_acc();

    // prologue: tile 0 -> buffer 0
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(i, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 fw[2][4], fx[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fw[0][i] = rdw(0, 0, i); fx[0][i] = rdx(0, 0, i); }

    // main loop: tile t in buffer t & 1; unrolled by two so that buffer indices are compile-time.  One other instruction
    // (a fragment read of the next sub-step, a copy of the next tile) after each MFMA: they issue while the MFMA runs.
    auto tile = [&](auto bufc, int t, bool has_next) {
        constexpr int B = decltype(bufc)::value;
        const int k1 = has_next ? (t + 1) * 64 : 0;      // (last tile: a harmless re-load of tile 0 instead of branches in the MFMA stream)
        asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(fw[0][0]), "v"(fx[0][0]));
        fw[1][0] = rdw(B, 1, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(fw[0][0]), "v"(fx[0][1]));
        fx[1][0] = rdx(B, 1, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(fw[0][0]), "v"(fx[0][2]));
        fw[1][1] = rdw(B, 1, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(fw[0][0]), "v"(fx[0][3]));
        fx[1][1] = rdx(B, 1, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" :: "v"(fw[0][1]), "v"(fx[0][0]));
        fw[1][2] = rdw(B, 1, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" :: "v"(fw[0][1]), "v"(fx[0][1]));
        fx[1][2] = rdx(B, 1, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" :: "v"(fw[0][1]), "v"(fx[0][2]));
        fw[1][3] = rdw(B, 1, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" :: "v"(fw[0][1]), "v"(fx[0][3]));
        fx[1][3] = rdx(B, 1, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" :: "v"(fw[0][2]), "v"(fx[0][0]));
        issue(0, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" :: "v"(fw[0][2]), "v"(fx[0][1]));
        issue(1, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" :: "v"(fw[0][2]), "v"(fx[0][2]));
        issue(2, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" :: "v"(fw[0][2]), "v"(fx[0][3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" :: "v"(fw[0][3]), "v"(fx[0][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" :: "v"(fw[0][3]), "v"(fx[0][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(fw[0][3]), "v"(fx[0][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(fw[0][3]), "v"(fx[0][3]));
        SB();
        asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(fw[1][0]), "v"(fx[1][0]));
        fw[0][0] = rdw(B, 2, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(fw[1][0]), "v"(fx[1][1]));
        fx[0][0] = rdx(B, 2, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(fw[1][0]), "v"(fx[1][2]));
        fw[0][1] = rdw(B, 2, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(fw[1][0]), "v"(fx[1][3]));
        fx[0][1] = rdx(B, 2, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" :: "v"(fw[1][1]), "v"(fx[1][0]));
        fw[0][2] = rdw(B, 2, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" :: "v"(fw[1][1]), "v"(fx[1][1]));
        fx[0][2] = rdx(B, 2, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" :: "v"(fw[1][1]), "v"(fx[1][2]));
        fw[0][3] = rdw(B, 2, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" :: "v"(fw[1][1]), "v"(fx[1][3]));
        fx[0][3] = rdx(B, 2, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" :: "v"(fw[1][2]), "v"(fx[1][0]));
        issue(3, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" :: "v"(fw[1][2]), "v"(fx[1][1]));
        issue(4, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" :: "v"(fw[1][2]), "v"(fx[1][2]));
        issue(5, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" :: "v"(fw[1][2]), "v"(fx[1][3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" :: "v"(fw[1][3]), "v"(fx[1][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" :: "v"(fw[1][3]), "v"(fx[1][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(fw[1][3]), "v"(fx[1][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(fw[1][3]), "v"(fx[1][3]));
        SB();
        asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(fw[0][0]), "v"(fx[0][0]));
        fw[1][0] = rdw(B, 3, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(fw[0][0]), "v"(fx[0][1]));
        fx[1][0] = rdx(B, 3, 0);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(fw[0][0]), "v"(fx[0][2]));
        fw[1][1] = rdw(B, 3, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(fw[0][0]), "v"(fx[0][3]));
        fx[1][1] = rdx(B, 3, 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" :: "v"(fw[0][1]), "v"(fx[0][0]));
        fw[1][2] = rdw(B, 3, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" :: "v"(fw[0][1]), "v"(fx[0][1]));
        fx[1][2] = rdx(B, 3, 2);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" :: "v"(fw[0][1]), "v"(fx[0][2]));
        fw[1][3] = rdw(B, 3, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" :: "v"(fw[0][1]), "v"(fx[0][3]));
        fx[1][3] = rdx(B, 3, 3);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" :: "v"(fw[0][2]), "v"(fx[0][0]));
        issue(6, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" :: "v"(fw[0][2]), "v"(fx[0][1]));
        issue(7, 1 - B, k1);
        asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" :: "v"(fw[0][2]), "v"(fx[0][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" :: "v"(fw[0][2]), "v"(fx[0][3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" :: "v"(fw[0][3]), "v"(fx[0][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" :: "v"(fw[0][3]), "v"(fx[0][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(fw[0][3]), "v"(fx[0][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(fw[0][3]), "v"(fx[0][3]));
        SB();
        asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(fw[1][0]), "v"(fx[1][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(fw[1][0]), "v"(fx[1][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(fw[1][0]), "v"(fx[1][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(fw[1][0]), "v"(fx[1][3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" :: "v"(fw[1][1]), "v"(fx[1][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" :: "v"(fw[1][1]), "v"(fx[1][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" :: "v"(fw[1][1]), "v"(fx[1][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" :: "v"(fw[1][1]), "v"(fx[1][3]));
        SB();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) { fw[0][i] = rdw(1 - B, 0, i); fx[0][i] = rdx(1 - B, 0, i); }
        SB();
        asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" :: "v"(fw[1][2]), "v"(fx[1][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" :: "v"(fw[1][2]), "v"(fx[1][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" :: "v"(fw[1][2]), "v"(fx[1][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" :: "v"(fw[1][2]), "v"(fx[1][3]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" :: "v"(fw[1][3]), "v"(fx[1][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" :: "v"(fw[1][3]), "v"(fx[1][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" :: "v"(fw[1][3]), "v"(fx[1][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" :: "v"(fw[1][3]), "v"(fx[1][3]));
        SB();
    };
    in_loop = true;
    for (int t = 0; t < nk; t += 2) {
        tile(std::integral_constant<int, 0>{}, t, t + 1 < nk);
        if (t + 1 < nk) tile(std::integral_constant<int, 1>{}, t + 1, t + 2 < nk);
    }

    // epilogue (probe: direct 8-byte stores of 4 consecutive couts per lane); accumulators read out 16 at a time
    asm volatile("s_nop 7\n s_nop 7" ::);
    float s_all = 0.f;
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a2" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a5" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a6" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a8" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a9" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a10" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a11" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a12" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a13" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a14" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a15" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 0 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 0 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a16" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a17" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a18" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a19" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a20" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a21" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a22" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a23" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a24" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a25" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a26" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a27" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a28" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a29" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a30" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a31" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 1 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 0 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a33" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a34" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a35" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a36" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a37" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a38" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a39" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a40" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a41" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a42" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a43" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a44" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a45" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a46" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a47" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 2 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 0 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a48" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a49" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a50" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a51" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a52" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a53" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a54" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a55" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a56" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a57" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a58" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a59" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a60" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a61" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a62" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a63" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 3 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 0 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a64" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a65" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a66" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a67" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a68" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a69" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a70" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a71" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a72" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a73" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a74" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a75" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a76" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a77" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a78" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a79" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 0 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 1 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a80" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a81" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a82" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a83" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a84" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a85" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a86" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a87" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a88" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a89" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a90" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a91" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a92" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a93" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a94" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a95" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 1 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 1 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a96" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a97" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a98" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a99" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a100" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a101" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a102" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a103" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a104" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a105" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a106" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a107" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a108" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a109" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a110" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a111" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 2 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 1 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a112" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a113" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a114" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a115" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a116" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a117" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a118" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a119" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a120" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a121" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a122" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a123" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a124" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a125" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a126" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a127" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 3 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 1 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a128" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a129" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a130" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a131" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a132" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a133" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a134" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a135" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a136" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a137" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a138" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a139" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a140" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a141" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a142" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a143" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 0 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 2 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a144" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a145" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a146" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a147" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a148" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a149" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a150" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a151" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a152" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a153" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a154" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a155" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a156" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a157" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a158" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a159" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 1 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 2 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a160" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a161" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a162" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a163" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a164" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a165" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a166" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a167" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a168" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a169" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a170" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a171" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a172" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a173" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a174" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a175" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 2 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 2 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a176" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a177" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a178" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a179" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a180" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a181" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a182" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a183" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a184" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a185" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a186" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a187" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a188" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a189" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a190" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a191" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 3 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 2 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a192" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a193" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a194" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a195" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a196" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a197" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a198" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a199" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a202" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a203" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a204" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a205" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a206" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a207" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 0 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 3 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a208" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a209" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a210" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a211" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a212" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a213" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a214" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a215" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a216" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a217" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a218" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a219" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a220" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a221" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a222" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a223" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 1 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 3 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a224" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a225" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a226" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a227" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a228" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a229" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a230" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a231" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a232" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a233" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a234" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a235" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a236" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a237" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a238" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a239" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 2 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 3 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    {
        float av[16];
        asm volatile("v_accvgpr_read_b32 %0, a240" : "=v"(av[0]));
        asm volatile("v_accvgpr_read_b32 %0, a241" : "=v"(av[1]));
        asm volatile("v_accvgpr_read_b32 %0, a242" : "=v"(av[2]));
        asm volatile("v_accvgpr_read_b32 %0, a243" : "=v"(av[3]));
        asm volatile("v_accvgpr_read_b32 %0, a244" : "=v"(av[4]));
        asm volatile("v_accvgpr_read_b32 %0, a245" : "=v"(av[5]));
        asm volatile("v_accvgpr_read_b32 %0, a246" : "=v"(av[6]));
        asm volatile("v_accvgpr_read_b32 %0, a247" : "=v"(av[7]));
        asm volatile("v_accvgpr_read_b32 %0, a248" : "=v"(av[8]));
        asm volatile("v_accvgpr_read_b32 %0, a249" : "=v"(av[9]));
        asm volatile("v_accvgpr_read_b32 %0, a250" : "=v"(av[10]));
        asm volatile("v_accvgpr_read_b32 %0, a251" : "=v"(av[11]));
        asm volatile("v_accvgpr_read_b32 %0, a252" : "=v"(av[12]));
        asm volatile("v_accvgpr_read_b32 %0, a253" : "=v"(av[13]));
        asm volatile("v_accvgpr_read_b32 %0, a254" : "=v"(av[14]));
        asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(av[15]));
        if (STORE) {
            const int m = m0 + wm * 128 + 3 * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + 3 * 32 + 8 * g + 4 * (lane >> 5);
                typedef __attribute__((ext_vector_type(2))) float f2;
                typedef __attribute__((ext_vector_type(2))) __bf16 b2;
                const f2 a = {av[g * 4 + 0], av[g * 4 + 1]}, b = {av[g * 4 + 2], av[g * 4 + 3]};
                uint2 v;
                v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, b2));
                v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, b2));
                *reinterpret_cast<uint2*>(C + (long)m * N + n) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all += av[r];
        }
    }
    if (!STORE && s_all == 12345.678f) C[0] = 1;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    struct { int M, N, K; const char* name; } shapes[] = {{524288, 256, 2304, "P2 3x3 (8x256x256, 256->256)"}, {8192, 1024, 12544, "fc1"},
                                                           {32768, 256, 2304, "layer3 3x3"}, {172800, 256, 2304, "FOA-sized"}};
    for (auto& sh : shapes) {
        const long M = sh.M, N = sh.N, K = sh.K;
        std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
        srand(1);
        for (auto& v : hx) v = f2bf((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : hw) v = f2bf((rand() % 2001 - 1000) / 8000.f);
        bf16_t *X, *W, *C;
        hipMalloc(&X, hx.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
        hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        const int grid = (int)((M / 256) * (N / 256));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                dim3 g(grid), b256(256);
                if (mode == 0) hipLaunchKernelGGL((gemm_quad_kernel<true, 0>), g, b256, 0, 0, X, W, C, (int)M, (int)N, (int)K);
                if (mode == 1) hipLaunchKernelGGL((gemm_quad_kernel<false, 0>), g, b256, 0, 0, X, W, C, (int)M, (int)N, (int)K);
                if (mode == 2) hipLaunchKernelGGL((gemm_quad_kernel<false, 1>), g, b256, 0, 0, X, W, C, (int)M, (int)N, (int)K);
                if (mode == 3) hipLaunchKernelGGL((gemm_quad_kernel<false, 2>), g, b256, 0, 0, X, W, C, (int)M, (int)N, (int)K);
                if (mode == 4) hipLaunchKernelGGL((gemm_quad_kernel<false, 3>), g, b256, 0, 0, X, W, C, (int)M, (int)N, (int)K);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const char* names[] = {"with stores", "no stores", "no stores, no fragment reads", "no stores, no copies", "MFMA stream only"};
            printf("%-32s %-30s: %.3f ms  %.0f TFLOP/s\n", sh.name, names[mode], best, 2.0 * M * N * K / best / 1e9);
        }
        // spot check 64 entries
        std::vector<uint16_t> hc((size_t)M * N);
        hipLaunchKernelGGL((gemm_quad_kernel<true, 0>), dim3(grid), dim3(256), 0, 0, X, W, C, (int)M, (int)N, (int)K);
        hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int s = 0; s < 64; ++s) {
            const long m = (long)rand() % M, n = rand() % N;
            double ref = 0;
            for (long k = 0; k < K; ++k) ref += (double)bf2f(hx[m * K + k]) * bf2f(hw[n * K + k]);
            const double d = fabs(ref - bf2f(hc[m * N + n])) / (fabs(ref) + 1e-2);
            if (d > worst) worst = d;
        }
        printf("   spot check: worst relative error %.4f\n", worst);
        hipFree(X); hipFree(W); hipFree(C);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error %s\n", hipGetErrorString(e));
    return 0;
}
