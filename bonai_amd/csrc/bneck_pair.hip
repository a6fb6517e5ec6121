// bneck_pair.hip -- two consecutive 1x1 convolutions of a ResNet stage as ONE launch, for gfx950 (loft_bneck_pair_bf16).
//
// Reference call sites (mmdet/models/backbones/resnet.py:266-298, Bottleneck.forward): block k ends with
//     out_k = relu(bn3(conv3(t2_k)) + identity)                     (256 -> 1024 channels in layer3, 128 -> 512 in layer2)
// and block k+1 begins with
//     t1_{k+1} = relu(bn1(conv1(out_k)))                            (1024 -> 256, 512 -> 128).
// As two launches out_k goes to HBM (67 MB in layer3 at 8 x 1024^2), comes back as conv1's operand (67 MB) and once more as block
// k+1's residual.  Here a workgroup keeps its 128-pixel slice of out_k in LDS, 128 channels at a time, between the two products:
//     for each chunk j of 128 channels:   mid_j  = f1(W1[j] . a_in + bias1[j] + res_j)   -> HBM once, and -> LDS
//                                         acc2  += W2[:, j] . mid_j                      (the second product's K axis = the chunk axis)
//     out2 = f2(acc2 + bias2)
// out_k is written once and read once (as the next block's residual); 235 -> 168 MB per pair in layer3.  The backward pass is the
// mirror image and runs through the same kernel (BWD): a_in = the gradient of t1_{k+1}, W1 = conv1_{k+1}'s data-gradient packing,
// res = the gradient arriving over block k+1's identity shortcut, mask1 = out_k (its ReLU), W2 = conv3_k's data-gradient packing,
// mask2 = t2_k: mid = the gradient of out_k (masked), out2 = the gradient of t2_k (masked).
//
// Rounding points are those of the separate launches (mid and out2 pass through the 16-bit type; fp32 accumulation; the second
// product reads the ROUNDED mid) and so is the order of the fp32 operations (K ascending in steps of 16, then + bias, then +
// residual -- the latter two as exact products on the matrix pipe): the outputs equal the separate launches' bit for bit on every
// tested shape at 256 planes; at 128 planes one element in ~5e5 is one unit of the 16-bit type apart (the final rounding of the
// matrix pipe's exact add against the VALU add of the separate launch's epilogue; tests/test_bneck_pair_gpu.py).
//
// Structure.  256 threads = 4 waves, ONE per SIMD with the whole 512-register file (no weight staging in LDS, no barrier in a K
// loop): wave w owns channels [32 w, +32) of every chunk in product 1 and channels [P/4 w, +P/4) of product 2, all 128 pixels.
// Weights come straight from L2 into registers as the MFMA's A operand, one product ahead: W2[:, j] is requested before product 1
// of chunk j starts, W1[j+1] before product 2 of chunk j (64 registers each: 1000+ cycles of MFMA between request and use).
// The residual chunk j+1 and the mask chunk j are requested a chunk ahead as row-contiguous 16-byte pieces; stores are
// fire-and-forget.  The pixel operand (B) is read from LDS with ds_read_b128: a_in tile [128][P], chunk buffers 2 x [128][128].
// LDS rows are a multiple of 256 B, 16-byte chunk c of row r at position c ^ (r & 15): fragment reads (32 consecutive rows, one
// chunk) and the epilogue's 8-byte read-modify-writes are conflict-free per 16 lanes.
// Weight layout "K8": [K/8][rows][8] (loft_pack_k8_multi re-arranges the [rows][K] packings once per step).  An A fragment = lane
// (row m, half g) <- 16 bytes at K block 2 step + g, row m: lanes 0-31 and 32-63 each read 512 CONTIGUOUS bytes.  From the
// [rows][K] packing the same fragment is 64 pieces of 16 bytes 512 bytes apart: the address unit then spent ~64 cycles per load
// instruction and the weight stream alone cost 19 of 65 us per layer3 pair (profiles/round6_probes/pair_time_v2.txt).  K runs in
// its natural order, 16 per MFMA step.
// Roofline: HBM (a_in + res + mid + out2 [+ mask1 + mask2]: 168 MB forward / 252 MB backward per layer3 pair at batch 8).
#include "conv_tap.h"
#include "../../include/loft_hip.h"

namespace {
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (a plain vector: arrays of HIP's uint4 STRUCT stayed in scratch)
struct PairArgs {
    const bf16_t* a_in;     // [M][P]
    const bf16_t* w1;       // K8 layout of [C][P] (rows = channels of mid, K = P): [P/8][C][8]
    const float* bias1;     // [C] | null
    const bf16_t* res;      // [M][C] | null
    const bf16_t* mask1;    // [M][C] | null   (BWD)
    bf16_t* mid;            // [M][C]
    const bf16_t* w2;       // K8 layout of [P][C] (rows = channels of out2, K = C): [C/8][P][8]
    const float* bias2;     // [P] | null
    const bf16_t* mask2;    // [M][P] | null   (BWD)
    bf16_t* out2;           // [M][P]
    long M;
    int C;
};

#define PAIR_SYNC()                                          \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                   \
        __builtin_amdgcn_s_barrier();                        \
        __builtin_amdgcn_sched_barrier(0);                   \
        asm volatile("" ::: "memory");                       \
    } while (0)

// ReLU of two packed 16-bit floats: a negative value has its sign bit set = is negative as an int16 (bfloat16 and binary16 alike)
typedef __attribute__((ext_vector_type(2))) short pair_s16x2;
__device__ __forceinline__ uint32_t pair_relu2(uint32_t w) {
    const pair_s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pair_s16x2, w), z));
}
__device__ __forceinline__ u32x4 pair_mask16(const u32x4 v, const u32x4 m) {
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float v0, v1, m0, m1;
        unpack2_16(v[d], v0, v1);
        unpack2_16(m[d], m0, m1);
        o[d] = pack2_bf16(m0 > 0.f ? v0 : 0.f, m1 > 0.f ? v1 : 0.f);
    }
    return o;
}

// ABL: timing ablations (loft_bneck_pair_bf16_v; results wrong): 1 no product 1, 2 no product 2, 4 no store of mid, 8 no residual
// re-load, 16 no epilogue 1, 32 weights loaded once
template <int P, bool BWD, int ABL = 0>
__global__ __launch_bounds__(256) void bneck_pair_kernel(const PairArgs a) {
    constexpr int ROWB = 2 * P;          // bytes per row of the a_in tile
    constexpr int CPR = ROWB / 16;       // 16-byte chunks per such row
    constexpr int KG1 = P / 64;          // 64-wide K groups of product 1
    constexpr int NT2 = P / 128;         // 32-channel A tiles per wave in product 2
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const opA = lds;
    char* const cbuf = lds + 128 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fq = lane >> 5;
    const long m0 = (long)blockIdx.x * 128;
    const int C = a.C, nch = C >> 7;

    // ---- prologue: the a_in tile, row-contiguous 16-byte pieces -> swizzled LDS rows
    {
        constexpr int NP = 128 * CPR / 256;
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = (i0 + i) * 256 + tid, r = pc / CPR, c = pc % CPR;
                v[i] = *reinterpret_cast<const u32x4*>(a.a_in + (m0 + r) * P + c * 8);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = (i0 + i) * 256 + tid, r = pc / CPR, c = pc % CPR;
                *reinterpret_cast<u32x4*>(opA + r * ROWB + ((c ^ (r & 15)) << 4)) = v[i];
            }
        }
    }
    // row-pass pieces of this thread inside a [128][128] chunk: piece i = row (i * 256 + tid) >> 4, chunk (tid & 15)
    const int pr0 = tid >> 4, pc0 = tid & 15;
    u32x4 rreg[8], mreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rreg[i] = *reinterpret_cast<const u32x4*>(a.res + (m0 + pr0 + 16 * i) * C + pc0 * 8);
    // weights of product 1, chunk 0
    bf16x8 w1r[KG1][4], w2r[NT2][2][4];
    {
        const bf16_t* wp = a.w1 + ((long)fq * C + 32 * wave + frow) * 8;
#pragma unroll
        for (int u = 0; u < KG1; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) w1r[u][s] = *reinterpret_cast<const bf16x8*>(wp + (long)(2 * (4 * u + s)) * C * 8);
    }
    // identity fragments (A operand, rows = the wave's 32 channels): step sr covers K = 16 sr + 8 fq + [0, 8)
    bf16x8 idA[2];
#pragma unroll
    for (int sr = 0; sr < 2; ++sr)
#pragma unroll
        for (int e = 0; e < 8; ++e) idA[sr][e] = (16 * sr + 8 * fq + e == frow) ? (short)LOFT_ONE16 : (short)0;
    bf16x8 ones3;                         // B operand of the bias step: K slots 0..2 of lanes 0-31 hold 1.0
#pragma unroll
    for (int e = 0; e < 8; ++e) ones3[e] = (e < 3 && fq == 0) ? (short)LOFT_ONE16 : (short)0;
    f32x16 acc2[NT2][4];
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][t][e] = 0.f;
    PAIR_SYNC();                          // the a_in tile is complete

    for (int j = 0; j < nch; ++j) {
        char* const buf = cbuf + (j & 1) * 32768;
        // (1) request the weights of product 2, chunk j -- BEFORE the HBM loads below: vmcnt retires in order, and a wait for these
        // L2 hits must not stand behind residual pieces that come from HBM
        if (!(ABL & 32) || j == 0) {
            const bf16_t* wp = a.w2 + ((long)(16 * j + fq) * P + 32 * NT2 * wave + frow) * 8;
#pragma unroll
            for (int i = 0; i < NT2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int s = 0; s < 4; ++s) w2r[i][u][s] = *reinterpret_cast<const bf16x8*>(wp + ((long)(2 * (4 * u + s)) * P + 32 * i) * 8);
        }
        // (2) residual chunk j -> LDS; request residual chunk j+1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = pr0 + 16 * i;
            *reinterpret_cast<u32x4*>(buf + r * 256 + ((pc0 ^ (r & 15)) << 4)) = rreg[i];
        }
        if constexpr (!(ABL & 8)) {
            const int jn = j + 1 < nch ? j + 1 : j;      // (last chunk: a harmless re-read instead of a branch around the array)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                rreg[i] = *reinterpret_cast<const u32x4*>(a.res + (m0 + pr0 + 16 * i) * C + 128 * jn + pc0 * 8);
        }
        float4 b1v[4];
#ifdef LOFT_ACT_F16
        if constexpr (!BWD) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) b1v[gq] = *reinterpret_cast<const float4*>(a.bias1 + 128 * j + 32 * wave + 8 * gq + 4 * fq);
        }
#endif
        // (3) product 1: mid chunk [32 channels of this wave][128 pixels], K = P
        f32x16 acc1[4], c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) c0[e] = 0.f;
        {
            // (one wave per SIMD: nothing else hides the LDS round trip, so step st + 1's fragments are requested before step st's MFMAs)
            bf16x8 bfr[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bfr[0][t] = *reinterpret_cast<const bf16x8*>(opA + (32 * t + frow) * ROWB + ((fq ^ (frow & 15)) << 4));
#pragma unroll
            for (int st = 0; st < 4 * KG1; ++st) {
                if (st + 1 < 4 * KG1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        bfr[(st + 1) & 1][t] =
                            *reinterpret_cast<const bf16x8*>(opA + (32 * t + frow) * ROWB + (((2 * (st + 1) + fq) ^ (frow & 15)) << 4));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (!(ABL & 1)) acc1[t] = LOFT_MFMA_32x32x16(w1r[st >> 2][st & 3], bfr[st & 1][t], st == 0 ? c0 : acc1[t]);
                    else if (st == 0) acc1[t] = c0;
                }
            }
        }
        // (4) request the weights of product 1, chunk j+1 (their registers are free now)
        if constexpr (!(ABL & 32)) {
            const int jn = j + 1 < nch ? j + 1 : j;
            const bf16_t* wp = a.w1 + ((long)fq * C + 128 * jn + 32 * wave + frow) * 8;
#pragma unroll
            for (int u = 0; u < KG1; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s) w1r[u][s] = *reinterpret_cast<const bf16x8*>(wp + (long)(2 * (4 * u + s)) * C * 8);
        }
        if constexpr (BWD) {              // mask chunk j: requested here (product 1's registers are at their peak before), used in (6)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                mreg[i] = *reinterpret_cast<const u32x4*>(a.mask1 + (m0 + pr0 + 16 * i) * C + 128 * j + pc0 * 8);
        }
        PAIR_SYNC();                      // B1: every thread's residual pieces of chunk j are in LDS
        // (5) the residual enters through the matrix pipe: acc += I . res (two K = 16 steps with identity fragments as the A operand, the
        // residual rows as B fragments: bf16 x 1.0 accumulated in fp32 is exact) -- 8 MFMAs instead of 16 LDS reads, 32 unpacks and
        // 64 adds per lane.  This wave alone reads and writes channels [32 wave, +32) of the chunk buffer: no barrier in between.
        if constexpr (!(ABL & 16)) {
            // (operation order of the separate launches' epilogue -- products, + bias, + residual, ReLU, one rounding -- so that the
            //  fused pair is BIT-IDENTICAL to them: the bias is added to the FINISHED sums.  bfloat16 build: on the matrix pipe as
            //  well -- the fp32 bias is the exact sum of three bfloat16 pieces (8 + 8 + 8 significant bits), one K step with
            //  A = [hi, mid, lo, 0 ..] per channel row and B = [1, 1, 1, 0 ..] adds it with a single rounding; as 64 VALU adds plus
            //  the accumulators' trip through the VALU registers it cost 6 of 61 us per layer3 pair.  binary16 build: VALU.)
            if constexpr (!BWD) {
#ifndef LOFT_ACT_F16
                const float b = a.bias1[128 * j + 32 * wave + frow];
                const float h0 = bf16_to_f32(f32_to_bf16(b)), r1 = b - h0, h1 = bf16_to_f32(f32_to_bf16(r1)), r2 = r1 - h1;
                bf16x8 bA;
#pragma unroll
                for (int e = 0; e < 8; ++e) bA[e] = 0;
                bA[0] = fq ? (short)0 : (short)f32_to_bf16(h0);
                bA[1] = fq ? (short)0 : (short)f32_to_bf16(h1);
                bA[2] = fq ? (short)0 : (short)f32_to_bf16(r2);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[t] = LOFT_MFMA_32x32x16(bA, ones3, acc1[t]);
#else
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        acc1[t][4 * gq] += b1v[gq].x; acc1[t][4 * gq + 1] += b1v[gq].y;
                        acc1[t][4 * gq + 2] += b1v[gq].z; acc1[t][4 * gq + 3] += b1v[gq].w;
                    }
#endif
            }
            bf16x8 rfr[2][4];
#pragma unroll
            for (int sr = 0; sr < 2; ++sr)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    rfr[sr][t] = *reinterpret_cast<const bf16x8*>(buf + (32 * t + frow) * 256 + (((4 * wave + 2 * sr + fq) ^ (frow & 15)) << 4));
#pragma unroll
            for (int sr = 0; sr < 2; ++sr)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[t] = LOFT_MFMA_32x32x16(idA[sr], rfr[sr][t], acc1[t]);
            // epilogue 1: (ReLU,) round to the 16-bit type, in place -- the B operand of product 2
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int px = 32 * t + frow;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    char* p = buf + px * 256 + (((4 * wave + gq) ^ (px & 15)) << 4) + 8 * fq;
                    uint2 w;
                    w.x = pack2_bf16(acc1[t][4 * gq], acc1[t][4 * gq + 1]);
                    w.y = pack2_bf16(acc1[t][4 * gq + 2], acc1[t][4 * gq + 3]);
                    if constexpr (!BWD) { w.x = pair_relu2(w.x); w.y = pair_relu2(w.y); }
                    *reinterpret_cast<uint2*>(p) = w;
                }
            }
        }
        PAIR_SYNC();                      // B2: the chunk is complete
        // (6) row pass: (mask,) store the chunk of mid, whole 256-byte runs per row
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = pr0 + 16 * i;
            char* p = buf + r * 256 + ((pc0 ^ (r & 15)) << 4);
            u32x4 v = *reinterpret_cast<const u32x4*>(p);
            if constexpr (BWD) {
                v = pair_mask16(v, mreg[i]);
                *reinterpret_cast<u32x4*>(p) = v;
            }
            if constexpr (!(ABL & 4)) *reinterpret_cast<u32x4*>(a.mid + (m0 + r) * C + 128 * j + pc0 * 8) = v;
            else if (v[0] == 0x12345678u) *reinterpret_cast<u32x4*>(a.mid + (m0 + r) * C + 128 * j + pc0 * 8) = v;
        }
        if constexpr (BWD) PAIR_SYNC();   // B3: product 2 reads the MASKED chunk
        // (7) product 2, K chunk j: acc2 += W2[:, 128 j .. +128) . mid_j
        {
            bf16x8 bfr[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bfr[0][t] = *reinterpret_cast<const bf16x8*>(buf + (32 * t + frow) * 256 + ((fq ^ (frow & 15)) << 4));
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        bfr[(st + 1) & 1][t] =
                            *reinterpret_cast<const bf16x8*>(buf + (32 * t + frow) * 256 + (((2 * (st + 1) + fq) ^ (frow & 15)) << 4));
                }
#pragma unroll
                for (int i = 0; i < NT2; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if constexpr (!(ABL & 2)) acc2[i][t] = LOFT_MFMA_32x32x16(w2r[i][st >> 2][st & 3], bfr[st & 1][t], acc2[i][t]);
            }
        }
        // (the other chunk buffer is written next; this one was last read two barriers ago by every wave)
    }

    // ---- epilogue 2: out2 tile [128][P] collected in the a_in tile's LDS (all its reads are behind B1 of the last chunk)
    PAIR_SYNC();
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
        const int ch0 = 32 * (NT2 * wave + i);
        float4 b2v[4];
        if constexpr (!BWD) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) b2v[gq] = *reinterpret_cast<const float4*>(a.bias2 + ch0 + 8 * gq + 4 * fq);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int px = 32 * t + frow;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc2[i][t][4 * gq + e];
                if constexpr (!BWD) {
                    v[0] += b2v[gq].x; v[1] += b2v[gq].y; v[2] += b2v[gq].z; v[3] += b2v[gq].w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                st4(reinterpret_cast<bf16_t*>(opA + px * ROWB + ((((ch0 >> 3) + gq) ^ (px & 15)) << 4) + 8 * fq), v);
            }
        }
    }
    PAIR_SYNC();
    {
        constexpr int NP = 128 * CPR / 256;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pc = i * 256 + tid, r = pc / CPR, c = pc % CPR;
            u32x4 v = *reinterpret_cast<const u32x4*>(opA + r * ROWB + ((c ^ (r & 15)) << 4));
            if constexpr (BWD) v = pair_mask16(v, *reinterpret_cast<const u32x4*>(a.mask2 + (m0 + r) * P + c * 8));
            *reinterpret_cast<u32x4*>(a.out2 + (m0 + r) * P + c * 8) = v;
        }
    }
}

// [R][K] -> [K/8][R][8], `n` matrices per launch (blockIdx.y): desc[i] = {src, dst, R, K}
__global__ __launch_bounds__(256) void pack_k8_multi_kernel(const long* __restrict__ desc) {
    const long* d = desc + 4 * blockIdx.y;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(d[0]);
    bf16_t* dst = reinterpret_cast<bf16_t*>(d[1]);
    const long R = d[2], K8 = d[3] >> 3;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < R * K8; i += (long)gridDim.x * 256) {
        const long kb = i / R, r = i - kb * R;             // consecutive threads: consecutive rows of one K block (contiguous 16-byte stores)
        *reinterpret_cast<u32x4*>(dst + i * 8) = *reinterpret_cast<const u32x4*>(src + (r * K8 + kb) * 8);
    }
}

template <int P, bool BWD, int ABL = 0>
int pair_launch(const PairArgs& pa, hipStream_t s) {
    const int lds_bytes = 128 * 2 * P + 65536;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)bneck_pair_kernel<P, BWD, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((bneck_pair_kernel<P, BWD, ABL>), dim3((unsigned)(pa.M / 128)), dim3(256), lds_bytes, s, pa);
    LOFT_LAUNCH_CHECK();
    return 0;
}
}  // namespace

LOFT_EXPORT int loft_pack_k8_multi(const int64_t* desc, int n, int64_t max_pieces, void* stream) {
    if (n <= 0 || max_pieces <= 0) return 0;
    long nb = (max_pieces + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(pack_k8_multi_kernel, dim3((unsigned)nb, (unsigned)n), dim3(256), 0, (hipStream_t)stream, (const long*)desc);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_bneck_pair_bf16_v(const void* a_in, const void* w1, const float* bias1, const void* res, const void* mask1, void* mid,
                                       const void* w2, const float* bias2, const void* mask2, void* out2, int64_t M, int P, int C, int variant,
                                       void* stream) {
    if (M <= 0) return 0;
    if (M % 128 || (P != 128 && P != 256) || C % 128 || C < 128) return (int)hipErrorInvalidValue;
    // forward form: both biases (and ReLUs), no mask; backward form: both masks, no bias.  The residual is part of both.
    const bool bwd = mask1 != nullptr;
    if (!res || (bwd ? (!mask2 || bias1 || bias2) : (mask2 || !bias1 || !bias2))) return (int)hipErrorInvalidValue;
    PairArgs pa;
    pa.a_in = (const bf16_t*)a_in; pa.w1 = (const bf16_t*)w1; pa.bias1 = bias1; pa.res = (const bf16_t*)res; pa.mask1 = (const bf16_t*)mask1;
    pa.mid = (bf16_t*)mid; pa.w2 = (const bf16_t*)w2; pa.bias2 = bias2; pa.mask2 = (const bf16_t*)mask2; pa.out2 = (bf16_t*)out2;
    pa.M = M; pa.C = C;
    hipStream_t s = (hipStream_t)stream;
    if (variant) {                    // timing ablations of the forward P = 256 instance (tools/probes/pair_time.py); results wrong
        if (P != 256 || bwd) return (int)hipErrorInvalidValue;
        switch (variant) {
            case 1: return pair_launch<256, false, 1>(pa, s);
            case 2: return pair_launch<256, false, 2>(pa, s);
            case 3: return pair_launch<256, false, 3>(pa, s);
            case 4: return pair_launch<256, false, 4>(pa, s);
            case 8: return pair_launch<256, false, 8>(pa, s);
            case 16: return pair_launch<256, false, 16>(pa, s);
            case 32: return pair_launch<256, false, 32>(pa, s);
            case 12: return pair_launch<256, false, 12>(pa, s);
            case 19: return pair_launch<256, false, 19>(pa, s);
            default: return (int)hipErrorInvalidValue;
        }
    }
    if (P == 256) return bwd ? pair_launch<256, true>(pa, s) : pair_launch<256, false>(pa, s);
    return bwd ? pair_launch<128, true>(pa, s) : pair_launch<128, false>(pa, s);
}
LOFT_EXPORT int loft_bneck_pair_bf16(const void* a_in, const void* w1, const float* bias1, const void* res, const void* mask1, void* mid,
                                     const void* w2, const float* bias2, const void* mask2, void* out2, int64_t M, int P, int C, void* stream) {
    return loft_bneck_pair_bf16_v(a_in, w1, bias1, res, mask1, mid, w2, bias2, mask2, out2, M, P, C, 0, stream);
}
