// conv_mfma.hip -- im2col-free NHWC "tap" convolution on gfx950 MFMA (bf16 in, fp32 accumulate).
//
// One kernel serves every dense contraction on the LOFT path
// (reference: nn.Conv2d/cuDNN call sites in mmdet/models/backbones/resnet.py:266-298,
//  necks/fpn.py:170-199, dense_heads/rpn_head.py:38-44, roi_heads/mask_heads/fcn_mask_head.py:118-126,
//  roi_heads/attribute_heads/offset_head_expand_feature.py:134-161, bbox_heads/convfc_bbox_head.py:135-173):
//
//   out[b, oy*os+oo_y, ox*os+oo_x, n] = act( bias[n] + res[...] +
//        sum_t sum_c src[b, oy*ss+dy[t], ox*ss+dx[t], c] * wgt[wt[t]][n][c] )
//
//   * forward conv RxS/stride s/pad p : taps (r-p, s-p), ss = s, os = 1
//   * data-gradient (dgrad), stride 1 : taps (p-r, p-s), weights packed transposed [t][Cin][Cout]
//   * dgrad of a stride-2 conv        : 4 output-parity classes, each its own tap subset (os = 2)
//   * ConvTranspose2d(k=2,s=2)        : 4 parity classes of one tap each (os = 2)
//   * nn.Linear                        : one tap, H = W = 1
//   * FOA's 4 rotation branches        : blockIdx.z = branch ("group") with per-group strides
//
// GEMM view: M = B*OH*OW output pixels, N = Cout, K = T*Cin.  The K loop walks (tap, 64-channel
// chunk); a chunk of one tap is a contiguous 128-byte run of the shifted source pixel, so the A
// tile is gathered straight from the NHWC activation with `global_load_lds` (16 B per lane, LDS
// destination lane-linear) -- no im2col buffer, out-of-image taps read a zero page.
//
// CDNA4 mapping: 256 threads = 4 wave64; block tile BMxBNx64, double-buffered in LDS (64 KiB ->
// 2 blocks/CU); v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as the A operand and the activation
// tile as the B operand so each lane ends up holding 4 consecutive output channels of one pixel
// (contiguous NHWC stores).  LDS rows are 128 B; 16-byte chunk q of row r is stored at slot
// q ^ ((r>>1)&7), applied on the per-lane *global source* address (the LDS image of a glds is
// lane-linear) and again on the ds_read_b128 address: conflict-free for the 16-lane groups that
// ds_read_b128 is serviced in.  Roofline: MFMA (dense bf16 2.5 PFLOP/s).
// Measured alternative (round 1): a 3-stage counted-vmcnt pipeline at 96 KiB LDS (1 block/CU) was SLOWER than this
// 2-stage / 2-blocks-per-CU form (fpn P2 3x3: 538 vs 649 TFLOP/s) -- the second resident block hides more than the
// deeper prefetch does; the next step is a 256-row, 8-wave tile with fragment double-buffering, not more stages.
#include "loft_common.h"
#include <algorithm>
#include <type_traits>
#include "../../include/loft_hip.h"
#include <stdlib.h>

#include "conv_tap.h"


// Coalesced copy of the [128 pixels][128 channels] bf16 tile of an output-shaped tensor (dense placement: pixel index == m)
// into LDS, 4 waves, swizzled as conv_epilogue expects.
__device__ __forceinline__ void conv_stage_tile(const ConvArgs& a, const bf16_t* t, long out_g, int m0, int n0, int wave, int lane,
                                                char* dst) {
    int aM = a.M, aCout = a.Cout;
    const bf16_t* zp = a.zero_page;
    LOFT_KEEP_S(aM); LOFT_KEEP_S(aCout);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 16 + wave * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ (row & 15);
        const int m = m0 + row, n = n0 + lc * 8;
        const bf16_t* p = (m < aM && n < aCout) ? t + out_g + (long)m * aCout + n : zp;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(dst + (it * 16 + wave * 4) * 256), 16, 0, 0);
    }
}

// The reverse: the finished [128][128] bf16 tile from LDS to the (dense) output, 16 bytes per lane, 16 lanes per 256-byte row.
__device__ __forceinline__ void conv_unstage_tile(const ConvArgs& a, long out_g, int m0, int n0, int wave, int lane, const char* srct) {
    int aM = a.M, aCout = a.Cout;
    void* aout = a.out;
    LOFT_KEEP_S(aM); LOFT_KEEP_S(aCout);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 16 + wave * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ (row & 15);
        const int m = m0 + row, n = n0 + lc * 8;
        if (m < aM && n < aCout) {
            const uint4 v = *reinterpret_cast<const uint4*>(srct + row * 256 + (lane & 15) * 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(aout) + out_g + (long)m * aCout + n) = v;
        }
    }
}

// STAGES = 2: double-buffered K loop.  STAGES = 1: single LDS buffer (half the LDS -> one more resident block per CU) for
// launches with only 1-2 K-steps (the K-shallow 1x1 convs), which are latency-bound: occupancy hides what a pipeline cannot.
// FAST: pixel-dependent address work hoisted out of the K loop (pays off from ~32 K-steps on; measured +9..12 % on the
// FOA / FC / layer4 shapes, -8 % on the 9-step layer1 3x3, so the dispatcher picks per launch).
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES = 2, bool FAST = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(STAGES == 1 ? 4 : 1)))
void conv_tap_kernel(const ConvArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;                 // 4 waves (128-wide tiles) or 8 waves (256x256 tile)
    constexpr int RPR = NW * 8;                           // tile rows staged per glds round (8 rows per wave)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_LOADS = BM / RPR, B_LOADS = BN / RPR;  // glds instructions per thread per K step
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    __shared__ __attribute__((aligned(16))) char lds[STAGES * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bz = V / (gridDim.x * gridDim.y), Vg = V - bz * (gridDim.x * gridDim.y);
    const int bx = a.nfast ? Vg / gridDim.y : Vg % gridDim.x, by = a.nfast ? Vg % gridDim.y : Vg / gridDim.x;
    const int m0 = bx * BM, n0 = by * BN;
    const int g = bz;
    const bf16_t* src = a.src + (long)g * a.src_gs;
    const bf16_t* wgt = a.wgt + (long)g * a.wgt_gs;

    const int lrow = lane >> 3, lchunk = lane & 7;
    const int ohw = a.OH * a.OW;
    const int kchunks = a.Cin / BK;
    const int nk = a.T * kchunks;

    // ---- weight rows this thread stages (both addressing schemes)
    int b_off[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int row = i * RPR + wave * 8 + lrow;
        const int n = n0 + row;
        b_off[i] = (n < a.Cout) ? (n * a.Cin + swz(row, lchunk) * 8) : -1;
    }

    // ---- activation rows.  Plain scheme (short K loops): pixel coordinates kept, address rebuilt per K-step.
    // FAST scheme (deep K loops): everything that depends on the pixel is computed ONCE -- the pointer of the row's un-shifted
    // source pixel and a bit mask of the taps that stay inside the map; a K-step then only adds a wave-uniform (scalar)
    // tap/channel offset and selects the zero page for masked taps: ~6 VALU per load instead of ~15 with two quarter-rate
    // integer multiplies and an exec-mask branch (the VALU shares its issue port with the MFMAs of the other wave).
    int a_base[A_LOADS], a_y[A_LOADS], a_x[A_LOADS], a_c[A_LOADS];
    const bf16_t* a_ptr[A_LOADS];
    unsigned a_mask[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = i * RPR + wave * 8 + lrow;
        const int m = m0 + row;
        a_c[i] = swz(row, lchunk) * 8;
        a_base[i] = 0; a_y[i] = -100000; a_x[i] = -100000;
        a_ptr[i] = src; a_mask[i] = 0u;
        if (m < a.M) {
            int b, rem;
            if (FAST && a.pixmajor) { rem = m / a.B; b = m - rem * a.B; } else { b = m / ohw; rem = m - b * ohw; }
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            a_base[i] = b * a.IH * a.IW;
            a_y[i] = oy * a.ss;
            a_x[i] = ox * a.ss;
            if constexpr (FAST) {
                a_ptr[i] = src + ((long)(a_base[i] + a_y[i] * a.IW + a_x[i]) * a.Cin + a_c[i]);
                unsigned msk = 0u;
                for (int t = 0; t < a.T; ++t) {
                    const int iy = a_y[i] + a.dy[t], ix = a_x[i] + a.dx[t];
                    msk |= ((iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW)) ? (1u << t) : 0u;
                }
                a_mask[i] = msk;
            }
        }
    }
    // pixmajor (RoI maps of 7x7 / 14x14 pixels, hundreds of RoIs): the 256 rows of a tile are ONE pixel position (at most two)
    // of 256 different RoIs, so a tap that leaves the map does so for the whole tile and its K-steps are skipped outright --
    // 18 % of the 3x3 taps of a 7x7 map, 9 % of a 14x14 one, are zero padding.  Memory layout and addresses are unchanged:
    // only the enumeration of the rows differs (every row is gathered / stored through its own pointer anyway).
    unsigned tmask = 0xffffffffu;
    int nk_eff = nk;
    if constexpr (FAST) {
        if (a.pixmajor) {
            // (scratch inside the ONE LDS array -- the last bytes of stage buffer 1, first overwritten by stage(1, 1) behind the K
            //  loop's first barrier.  A second __shared__ object makes hipcc treat every global_load_lds as a possible write to
            //  whatever a later ds_read touches: it then put `s_waitcnt vmcnt(0)` between stage() and compute(), i.e. the
            //  prefetch of K-step k+1 was waited for BEFORE computing K-step k -- found in the ISA of round 1's kernel.)
            static_assert(STAGES == 2, "FAST variants are double-buffered");
            unsigned* wor = reinterpret_cast<unsigned*>(lds + sizeof(lds) - 64);
            unsigned mm = 0u;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) mm |= a_mask[i];
            for (int o = 32; o > 0; o >>= 1) mm |= (unsigned)__shfl_xor((int)mm, o, 64);
            if (lane == 0) wor[wave] = mm;
            __syncthreads();
            tmask = 0u;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) tmask |= wor[w2];
            nk_eff = __popc(tmask) * kchunks;
        }
    }
    // FAST: running (wave-uniform) position of the NEXT K-step to stage
    int st_t = 0, st_c = 0;
    if constexpr (FAST) { while (st_t < a.T - 1 && !((tmask >> st_t) & 1u)) ++st_t; }
    long st_aoff = ((long)a.dy[st_t] * a.IW + a.dx[st_t]) * a.Cin;   // element offset of tap st_t relative to the base pixel
    const bf16_t* st_w = wgt + (long)a.wt[st_t] * a.Cout * a.Cin;

    auto stage = [&](int kk, int buf) {
        char* abuf = lds + buf * (A_BYTES + B_BYTES);
        char* bbuf = abuf + A_BYTES;
        const bf16_t* wt;
        if constexpr (FAST) {
            const long aoff = st_aoff + st_c;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const bf16_t* p = ((a_mask[i] >> st_t) & 1u) ? a_ptr[i] + aoff : a.zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(abuf + (i * RPR + wave * 8) * 128), 16, 0, 0);
            }
            wt = st_w + st_c;
            st_c += BK;
            if (st_c == a.Cin) {
                st_c = 0;
                ++st_t;
                while (st_t < a.T && !((tmask >> st_t) & 1u)) ++st_t;      // (pixmajor: taps outside the map for the whole tile)
                if (st_t < a.T) {
                    st_aoff = ((long)a.dy[st_t] * a.IW + a.dx[st_t]) * a.Cin;
                    st_w = wgt + (long)a.wt[st_t] * a.Cout * a.Cin;
                }
            }
        } else {
            const int t = kk / kchunks, c0 = (kk - t * kchunks) * BK;
            const int dy = a.dy[t], dx = a.dx[t];
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int iy = a_y[i] + dy, ix = a_x[i] + dx;
                const bool ok = (iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW);
                const bf16_t* p = ok ? src + ((long)(a_base[i] + iy * a.IW + ix) * a.Cin + c0 + a_c[i]) : a.zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(abuf + (i * RPR + wave * 8) * 128), 16, 0, 0);
            }
            wt = wgt + (long)a.wt[t] * a.Cout * a.Cin + c0;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const bf16_t* p = (b_off[i] >= 0) ? wt + b_off[i] : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(bbuf + (i * RPR + wave * 8) * 128), 16, 0, 0);
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frow = lane & 31, fq = lane >> 5;

    // Fragment addresses are loop invariant: row r+32 has the same swizzle as row r, so the NT / MT fragments of a sub-step sit
    // at fixed 4 KiB strides (ds_read immediate offsets) from one per-(buffer, sub-step) base.  The 16 bases live in VGPRs and
    // the K loop is unrolled by two so the buffer index is a compile-time constant: no address VALU between the MFMAs
    // (the SQ counters showed ~5 VALU instructions per MFMA before, mostly this address math, competing for the issue port).
    const char* wbase[STAGES][4];
    const char* xbase[STAGES][4];
    {
        const int rw = wn * WN + frow, rx = wm * WM + frow;
#pragma unroll
        for (int bsel = 0; bsel < STAGES; ++bsel)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int q = ks * 2 + fq;
                xbase[bsel][ks] = lds + bsel * (A_BYTES + B_BYTES) + rx * 128 + swz(rx, q) * 16;
                wbase[bsel][ks] = lds + bsel * (A_BYTES + B_BYTES) + A_BYTES + rw * 128 + swz(rw, q) * 16;
            }
    }
    auto compute = [&](auto bufc) {
        constexpr int bsel = decltype(bufc)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 wf[NT], xf[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wbase[bsel][ks] + i * 4096);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xbase[bsel][ks] + j * 4096);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = LOFT_MFMA_32x32x16(wf[i], xf[j], acc[i][j]);
        }
    };
    using buf0_t = std::integral_constant<int, 0>;
    using buf1_t = std::integral_constant<int, STAGES - 1>;

    stage(0, 0);
    if constexpr (STAGES == 2) {
        for (int kk = 0; kk < nk_eff; kk += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kk + 1 < nk_eff) stage(kk + 1, 1);
            compute(buf0_t{});
            if (kk + 1 < nk_eff) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kk + 2 < nk_eff) stage(kk + 2, 0);
                compute(buf1_t{});
            }
        }
    } else {
        for (int kk = 0; kk < nk_eff; ++kk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(buf0_t{});
            if (kk + 1 < nk_eff) {
                __syncthreads();          // everyone is done reading the single buffer
                stage(kk + 1, 0);
            }
        }
    }

    if constexpr (BM == 128 && BN == 128) {
        // memory-bound launches (residual add of a bottleneck's conv3, shortcut + ReLU mask of a fused block's first dgrad):
        // bring the residual / mask tiles in through the (now idle) stage buffers with coalesced copies
        const bool dense = a.os == 1 && a.OHf == a.OH && a.OWf == a.OW && !a.out_f32;
        // (the single-stage form has room for ONE tile: the dispatcher never gives it a residual AND a mask)
        if (dense && (STAGES == 2 ? (a.residual || a.mask || a.staged_out) : !(a.residual && a.mask))) {
            __syncthreads();                                    // all fragment reads of the last K-step are done
            const long out_g = (long)g * a.out_gs;
            char* rt = lds;
            char* mt = (STAGES == 2 || a.residual) ? lds + 32768 : lds;
            if (a.residual) conv_stage_tile(a, a.residual, out_g, m0, n0, wave, lane, rt);
            if (a.mask) conv_stage_tile(a, a.mask, out_g, m0, n0, wave, lane, mt);
            if (a.residual || a.mask) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // staged_out: the results replace the residual tile in place (each lane rewrites exactly the 8 bytes it read) and
            // leave as 16-byte-per-lane row-contiguous stores instead of 8-byte stores to 32 different rows per instruction
            char* ot = (a.mask && !a.residual) ? mt : rt;
            conv_epilogue<NT, MT, WM, WN>(a, acc, g, m0, n0, wm, wn, frow, fq, ohw, a.residual ? rt : nullptr,
                                          a.mask ? mt : nullptr, a.staged_out ? ot : nullptr);
            if (a.staged_out) {
                __syncthreads();
                conv_unstage_tile(a, out_g, m0, n0, wave, lane, ot);
            }
            return;
        }
    }
    conv_epilogue<NT, MT, WM, WN>(a, acc, g, m0, n0, wm, wn, frow, fq, ohw, nullptr, nullptr, nullptr, FAST && a.pixmajor);
}


// =====================================================================================
// 64 -> 64 channel stride-1 convs at high resolution (HRNet's first two branches, hrnet.py via resnet.py:13-92 BasicBlock; the
// 32-channel branch is carried zero-padded at 64).  The generic tap kernel gives such a layer a K-step of 8 MFMAs per wave per
// global->LDS round trip (9 round trips per 128-pixel tile: 110 us for 8 x 256^2 pixels, 1.8 TB/s).  Here the packed weights of
// ALL taps (T x 64 x 64 bf16 = 72 KiB) stay in LDS for the lifetime of a persistent workgroup, which walks 16 x 16-PIXEL PATCHES:
// the 18 x 18 halo of a 16 x 16 patch (41 KiB, double-buffered: 154 KiB of LDS in all, one workgroup per CU) is staged once and
// every tap reads its shifted rows from it -- one round trip per 72 MFMAs per wave, hidden behind the previous patch.
// 8 waves: wave w -> patch rows 2w, 2w+1 (32 pixels) x all 64 output channels.  128-byte LDS rows, 16-byte chunk q of row r at
// q ^ ((r >> 1) & 7) (round 6: with q ^ (r & 7) the sixteen rows of a ds_read_b128 lane group hit eight slots twice).
// Epilogue as the tap kernels: bias, residual, ReLU, ReLU-backward mask, bf16 store (8 bytes per lane).
// =====================================================================================
struct Conv64Args {
    const bf16_t* src; const bf16_t* wgt; const float* bias; const bf16_t* residual; const bf16_t* mask; bf16_t* out;
    const bf16_t* zero_page;
    int B, H, W, T, relu;
    int dy[9], dx[9], wt[9];
};

__global__ __launch_bounds__(512) void conv64_patch_kernel(const Conv64Args a, int ptx, int pty, int npatch) {
    constexpr int HW_ = 18, HR = 18 * 18;              // halo of a 16 x 16 patch
    constexpr int SLOTS = 41, HB = SLOTS * 8 * 128;    // 41 wave-copies of 8 rows (328 >= 324)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;                                    // [T * 64 rows][128 B]
    char* hl = lds + a.T * 64 * 128;                   // 2 x halo
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, pc = lane & 7;
    // ---- all taps' weights, once
    for (int r8 = wave; r8 < a.T * 8; r8 += 8) {       // 8 rows per wave-level copy
        const int row = r8 * 8 + lrow;
        const int q = pc ^ ((row >> 1) & 7);      // (16 consecutive rows -> 16 different 16-byte slots of the 256-byte bank row)
        __builtin_amdgcn_global_load_lds((gptr_t)(a.wgt + (long)row * 64 + q * 8), (lds_ptr_t)(wl + r8 * 8 * 128), 16, 0, 0);
    }
    auto stage = [&](int p, int buf) {
        char* hb = hl + buf * HB;
        const int b = p / (ptx * pty), rem = p - b * (ptx * pty);
        const int y0 = (rem / ptx) * 16, x0 = (rem % ptx) * 16;
        for (int slot = wave; slot < SLOTS; slot += 8) {
            const int row = slot * 8 + lrow;
            const int q = pc ^ ((row >> 1) & 7);      // (16 consecutive rows -> 16 different 16-byte slots of the 256-byte bank row)
            const int hy = row / HW_, hx = row - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bf16_t* ptr = a.zero_page;
            if (row < HR && y >= 0 && y < a.H && x >= 0 && x < a.W) ptr = a.src + ((long)(b * a.H + y) * a.W + x) * 64 + q * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lds_ptr_t)(hb + slot * 8 * 128), 16, 0, 0);
        }
    };
    const int frow = lane & 31, fq = lane >> 5;
    const int ppy = 2 * wave + (frow >> 4), ppx = frow & 15;    // this lane's pixel inside the patch (wave w: patch rows 2w, 2w+1)
    int p = blockIdx.x;
    if (p < npatch) stage(p, 0);
    for (int s2 = 0; p < npatch; p += gridDim.x, ++s2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p + (int)gridDim.x < npatch) stage(p + gridDim.x, (s2 + 1) & 1);
        const char* hb = hl + (s2 & 1) * HB;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // Fragments of tap t + 1 are requested BEFORE the MFMAs of tap t (round 6): in the straight form -- read, wait, multiply --
        // hipcc waited `lgkmcnt(0)` in front of nearly every MFMA (98 waits for 72 MFMAs in the ISA), i.e. an LDS round trip of
        // ~120 cycles per 32-cycle MFMA: 72 us per 8 x 256^2 launch against 18 us of matrix time and a 22 us HBM floor.  Twelve
        // 16-byte fragments per tap and set (4 activation + 8 weight), two sets; same MFMA order: bit-identical.
        bf16x8 xfr[2][4], wfr[2][8];
        auto rd_tap = [&](int t, auto setc) {
            constexpr int S_ = decltype(setc)::value;
            const int hr = (ppy + 1 + a.dy[t]) * HW_ + ppx + 1 + a.dx[t];
            const int wr = a.wt[t] * 64 + frow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int q = ks * 2 + fq;
                xfr[S_][ks] = *reinterpret_cast<const bf16x8*>(hb + hr * 128 + ((q ^ ((hr >> 1) & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int wri = wr + i * 32;
                    wfr[S_][ks * 2 + i] = *reinterpret_cast<const bf16x8*>(wl + wri * 128 + ((q ^ ((wri >> 1) & 7)) << 4));
                }
            }
        };
        auto mm_tap = [&](auto setc) {
            constexpr int S_ = decltype(setc)::value;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = LOFT_MFMA_32x32x16(wfr[S_][ks * 2 + i], xfr[S_][ks], acc[i]);
        };
        using s0_t = std::integral_constant<int, 0>;
        using s1_t = std::integral_constant<int, 1>;
        rd_tap(0, s0_t{});
#pragma unroll
        for (int t = 0; t < 9; t += 2) {
            if (t < a.T) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < a.T) rd_tap(t + 1, s1_t{});
                __builtin_amdgcn_sched_barrier(0);
                mm_tap(s0_t{});
            }
            if (t + 1 < a.T) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < a.T) rd_tap(t + 2, s0_t{});
                __builtin_amdgcn_sched_barrier(0);
                mm_tap(s1_t{});
            }
        }
        // ---- epilogue (round 6): the 256-pixel x 64-channel result is collected in the LDS rows of the halo it was computed from and
        // leaves as whole 128-byte pixel rows (16 bytes per lane, 8 rows per wave-level store); the residual tile comes in the same
        // way (global->LDS copies of whole rows), every lane adds its 8-byte pieces in fp32 and overwrites them with the result;
        // the ReLU-backward mask is applied to the packed 16-bit values in the copy-out pass against 16-byte mask loads.  The direct
        // form -- 8 bytes per lane to 32 different pixel rows per instruction, residual / mask read the same way -- cost more than
        // the patch's 72 MFMAs per wave (config 5: 264 launches, 72.7 us each against a 22 us HBM floor).  Same operations in the
        // same order on the same values: bit-identical to the direct form.
        const int b = p / (ptx * pty), rem = p - b * (ptx * pty);
        const int y0p = (rem / ptx) * 16, x0p = (rem % ptx) * 16;
        char* tb = const_cast<char*>(hb);               // tile [256 pixels][128 B]; 16-byte chunk c of row r at c ^ (r & 7)
        __syncthreads();                                // every wave is done with this patch's halo rows
        if (a.residual) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (wave * 4 + k) * 8 + lrow;
                const int y = y0p + (row >> 4), x = x0p + (row & 15);
                const bf16_t* ptr = a.zero_page;
                if (y < a.H && x < a.W) ptr = a.residual + ((long)(b * a.H + y) * a.W + x) * 64 + (pc ^ (row & 7)) * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lds_ptr_t)(tb + (wave * 4 + k) * 8 * 128), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        {
            const int r = ppy * 16 + ppx;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = i * 32 + 8 * gq + 4 * fq;
                    bf16_t* q_ = reinterpret_cast<bf16_t*>(tb + r * 128 + (((i * 4 + gq) ^ (r & 7)) << 4) + fq * 8);
                    float v[4] = {acc[i][gq * 4 + 0], acc[i][gq * 4 + 1], acc[i][gq * 4 + 2], acc[i][gq * 4 + 3]};
                    if (a.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
                        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    }
                    if (a.residual) {
                        float rv[4];
                        ld4(q_, rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    st4(q_, v);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        {
            // keep a 16-bit lane where its mask value is > 0: sign bit clear and magnitude bits non-zero
            auto sel = [](unsigned vv, unsigned mm) {
                const unsigned lo16 = ((mm & 0x8000u) == 0u && (mm & 0x7fffu) != 0u) ? 0xffffu : 0u;
                const unsigned hi16 = ((mm & 0x80000000u) == 0u && (mm & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
                return vv & (lo16 | hi16);
            };
            uint4 val[4], mk[4];
            long off[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (wave * 4 + k) * 8 + lrow;
                const int y = y0p + (row >> 4), x = x0p + (row & 15);
                ok[k] = y < a.H && x < a.W;
                off[k] = ((long)(b * a.H + y) * a.W + x) * 64 + (pc ^ (row & 7)) * 8;
                val[k] = *reinterpret_cast<const uint4*>(tb + row * 128 + pc * 16);
                if (a.mask && ok[k]) mk[k] = *reinterpret_cast<const uint4*>(a.mask + off[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!ok[k]) continue;
                uint4 o = val[k];
                if (a.mask) { o.x = sel(o.x, mk[k].x); o.y = sel(o.y, mk[k].y); o.z = sel(o.z, mk[k].z); o.w = sel(o.w, mk[k].w); }
                *reinterpret_cast<uint4*>(a.out + off[k]) = o;
            }
        }
    }
}

// =====================================================================================
// Tail of a 64-plane ResNet bottleneck in ONE launch (the frozen layer1 of the LOFT backbone, resnet.py:266-298; inference of
// any such block):   out = relu( W3 . relu(conv3x3(t1) + b2) + b3 + shortcut ),   t1 = the block's first 1x1 conv output.
// The three launches it replaces move 64 + 64 + 256 (+ 256 + 256 for a conv shortcut) channels per pixel through HBM at 137 .. 580
// TFLOP/s (K = 64: 1.7 FLOP per byte per channel -- "Why the conv family does not move further", round 3); fused, the 3x3's
// output never leaves the CU: conv64_patch_kernel's persistent patch walk (all nine taps' weights resident in LDS, 18 x 18 halo
// of t1 double-buffered), then the 64-channel result goes -- bias, ReLU, bf16 -- into the LDS space of the halo it came from and
// is the B operand of the 1x1 expansion, whose weights (a wave's 32 output channels x 64: 16 VGPRs) live in registers.
//   DS = false: shortcut = the block input (256 channels), read in the epilogue like any residual;
//   DS = true (the first block): shortcut = Wd . x + bd (x: the 64-channel block input): a second accumulator fed with x fragments
//     read straight from global memory in operand layout (16 bytes per lane), rounded to the 16-bit type before the add.  The
//     separate shortcut launch, its 256-channel output and the residual read of it disappear.
// Rounding points and MFMA order are those of the unfused launches (and of the oracle's 16-bit-points mode): bit-identical output.
// =====================================================================================
struct BneckArgs {
    Conv64Args c;                 // src = t1, wgt = W2 [9][64][64], bias = b2, out = out [B,H,W,256]; residual / mask unused
    const bf16_t* w3;             // [256][64]
    const float* b3;              // [256]
    const bf16_t* sc;             // DS ? x [B,H,W,64] : identity [B,H,W,256]
    const bf16_t* wd;             // DS: [256][64]
    const float* bd;              // DS: [256]
};

template <bool DS>
__global__ __launch_bounds__(512) void bneck_tail_kernel(const BneckArgs ba, int ptx, int pty, int npatch) {
    const Conv64Args& a = ba.c;
    constexpr int HW_ = 18, HR = 18 * 18;
    constexpr int SLOTS = 41, HB = SLOTS * 8 * 128;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;                                    // [9 * 64 rows][128 B]
    char* hl = lds + 9 * 64 * 128;                     // 2 x halo (the current one becomes the t2 tile [256 px][128 B])
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, pc = lane & 7;
    for (int r8 = wave; r8 < 9 * 8; r8 += 8) {
        const int row = r8 * 8 + lrow;
        const int q = pc ^ (row & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(a.wgt + (long)row * 64 + q * 8), (lds_ptr_t)(wl + r8 * 8 * 128), 16, 0, 0);
    }
    auto stage = [&](int p, int buf) {
        char* hb = hl + buf * HB;
        const int b = p / (ptx * pty), rem = p - b * (ptx * pty);
        const int y0 = (rem / ptx) * 16, x0 = (rem % ptx) * 16;
        for (int slot = wave; slot < SLOTS; slot += 8) {
            const int row = slot * 8 + lrow;
            const int q = pc ^ (row & 7);
            const int hy = row / HW_, hx = row - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bf16_t* ptr = a.zero_page;
            if (row < HR && y >= 0 && y < a.H && x >= 0 && x < a.W) ptr = a.src + ((long)(b * a.H + y) * a.W + x) * 64 + q * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lds_ptr_t)(hb + slot * 8 * 128), 16, 0, 0);
        }
    };
    const int frow = lane & 31, fq = lane >> 5;
    const int ppy = 2 * wave + (frow >> 4), ppx = frow & 15;
    // the 1x1 weights of this wave's 32 output channels, and their biases (lane: channels 32 wave + 8 gq + 4 fq + 0..3)
    bf16x8 w3r[4], wdr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        w3r[ks] = *reinterpret_cast<const bf16x8*>(ba.w3 + (long)(32 * wave + frow) * 64 + ks * 16 + fq * 8);
        if constexpr (DS) wdr[ks] = *reinterpret_cast<const bf16x8*>(ba.wd + (long)(32 * wave + frow) * 64 + ks * 16 + fq * 8);
    }
    // both bias vectors live in the 6 KiB of LDS behind the halos (registers are needed for the shortcut rows in flight)
    float* bl = reinterpret_cast<float*>(hl + 2 * HB);             // [64] b2, [256] b3, DS: [256] bd
    if (tid < 64) bl[tid] = a.bias ? a.bias[tid] : 0.f;
    if (tid < 256) bl[64 + tid] = ba.b3[tid];
    if (DS && tid < 256) bl[320 + tid] = ba.bd[tid];
    int p = blockIdx.x;
    if (p < npatch) stage(p, 0);
    for (int s2 = 0; p < npatch; p += gridDim.x, ++s2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // this patch's halo landed; the previous patch's t2 reads are done
        if (p + (int)gridDim.x < npatch) stage(p + gridDim.x, (s2 + 1) & 1);
        char* hb = hl + (s2 & 1) * HB;
        const int b = p / (ptx * pty), rem = p - b * (ptx * pty);
        const int py0 = (rem / ptx) * 16, px0 = (rem % ptx) * 16;
        // identity shortcut: this lane's 8 x 4 residual pieces of the patch are requested NOW and consumed in the expansion's
        // epilogue, behind the whole 3x3 (requested there, every pixel tile waited out an L2 / HBM round trip of its own: the
        // fused identity blocks ran no faster than the launches they replace)
        uint2 idr[DS ? 1 : 8][4];
        if constexpr (!DS) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int r = mt * 32 + frow;
                const int y = py0 + (r >> 4), x = px0 + (r & 15);
                const bool in = y < a.H && x < a.W;
                const bf16_t* sp = in ? ba.sc + ((long)(b * a.H + y) * a.W + x) * 256 + 32 * wave + 4 * fq : a.zero_page;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) idr[mt][gq] = *reinterpret_cast<const uint2*>(sp + (in ? 8 * gq : 0));
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hr = (ppy + 1 + a.dy[t]) * HW_ + ppx + 1 + a.dx[t];
            const int wr = a.wt[t] * 64 + frow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int q = ks * 2 + fq;
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(hb + hr * 128 + ((q ^ (hr & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int wri = wr + i * 32;
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + wri * 128 + ((q ^ (wri & 7)) << 4));
                    acc[i] = LOFT_MFMA_32x32x16(wf, xf, acc[i]);
                }
            }
        }
        __syncthreads();                               // every wave is done with the halo: it becomes the t2 tile
        {
            const int r = 32 * wave + frow;            // t2 row of this lane's pixel
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bv = *reinterpret_cast<const float4*>(bl + i * 32 + 8 * gq + 4 * fq);
                    float v[4] = {fmaxf(acc[i][gq * 4 + 0] + bv.x, 0.f), fmaxf(acc[i][gq * 4 + 1] + bv.y, 0.f),
                                  fmaxf(acc[i][gq * 4 + 2] + bv.z, 0.f), fmaxf(acc[i][gq * 4 + 3] + bv.w, 0.f)};
                    uint2 pk;
                    pk.x = pack2_bf16(v[0], v[1]); pk.y = pack2_bf16(v[2], v[3]);
                    const int q = i * 4 + gq;
                    *reinterpret_cast<uint2*>(hb + r * 128 + ((q ^ (r & 7)) << 4) + 8 * fq) = pk;
                }
        }
        __syncthreads();
        bf16x8 xs[2][4];                              // DS: x fragments of the pixel tile after the current one
        auto fetch_x = [&](int mt, bf16x8 (&dst)[4]) {
            const int r = mt * 32 + frow;
            const int y = py0 + (r >> 4), x = px0 + (r & 15);
            const bf16_t* xp = (y < a.H && x < a.W) ? ba.sc + ((long)(b * a.H + y) * a.W + x) * 64 : a.zero_page;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(xp + ks * 16 + fq * 8);
        };
        if constexpr (DS) fetch_x(0, xs[0]);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int r = mt * 32 + frow;
            const int y = py0 + (r >> 4), x = px0 + (r & 15);
            const bool in = y < a.H && x < a.W;
            const long pix = ((long)(b * a.H + y) * a.W + x);
            f32x16 c3, cd;
#pragma unroll
            for (int e = 0; e < 16; ++e) { c3[e] = 0.f; cd[e] = 0.f; }
            if constexpr (DS) {
                if (mt + 1 < 8) fetch_x(mt + 1, xs[(mt + 1) & 1]);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int q = ks * 2 + fq;
                const bf16x8 tf = *reinterpret_cast<const bf16x8*>(hb + r * 128 + ((q ^ (r & 7)) << 4));
                c3 = LOFT_MFMA_32x32x16(w3r[ks], tf, c3);
            }
            if constexpr (DS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) cd = LOFT_MFMA_32x32x16(wdr[ks], xs[mt & 1][ks], cd);
            }
            if (in) {
                const long o0 = pix * 256 + 32 * wave + 4 * fq;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bv = *reinterpret_cast<const float4*>(bl + 64 + 32 * wave + 8 * gq + 4 * fq);
                    float v[4] = {c3[gq * 4 + 0] + bv.x, c3[gq * 4 + 1] + bv.y, c3[gq * 4 + 2] + bv.z, c3[gq * 4 + 3] + bv.w};
                    float rv[4];
                    if constexpr (!DS) {
                        unpack2_16(idr[mt][gq].x, rv[0], rv[1]); unpack2_16(idr[mt][gq].y, rv[2], rv[3]);
                    } else {
                        // the shortcut conv's output passes through the 16-bit type, as it does between the unfused launches and
                        // in the oracle's 16-bit-points mode: same rounding points, bit-identical block output
                        const float4 dv = *reinterpret_cast<const float4*>(bl + 320 + 32 * wave + 8 * gq + 4 * fq);
                        unpack2_16(pack2_bf16(cd[gq * 4 + 0] + dv.x, cd[gq * 4 + 1] + dv.y), rv[0], rv[1]);
                        unpack2_16(pack2_bf16(cd[gq * 4 + 2] + dv.z, cd[gq * 4 + 3] + dv.w), rv[2], rv[3]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    st4(a.out + o0 + 8 * gq, v);
                }
            }
        }
    }
}

LOFT_EXPORT int loft_bneck_tail_bf16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* shortcut,
                                     const void* wd, const float* bd, void* out, const void* zero_page, int B, int H, int W, const int* dy_host,
                                     const int* dx_host, const int* wt_host, void* stream) {
    if (B < 1 || H < 1 || W < 1 || (wd && !bd)) return (int)hipErrorInvalidValue;
    BneckArgs ba;
    ba.c.src = (const bf16_t*)t1; ba.c.wgt = (const bf16_t*)w2; ba.c.bias = b2; ba.c.residual = nullptr; ba.c.mask = nullptr;
    ba.c.out = (bf16_t*)out; ba.c.zero_page = (const bf16_t*)zero_page; ba.c.B = B; ba.c.H = H; ba.c.W = W; ba.c.T = 9; ba.c.relu = 1;
    for (int t = 0; t < 9; ++t) {
        if (dy_host[t] < -1 || dy_host[t] > 1 || dx_host[t] < -1 || dx_host[t] > 1 || wt_host[t] < 0 || wt_host[t] > 8) return (int)hipErrorInvalidValue;
        ba.c.dy[t] = dy_host[t]; ba.c.dx[t] = dx_host[t]; ba.c.wt[t] = wt_host[t];
    }
    ba.w3 = (const bf16_t*)w3; ba.b3 = b3; ba.sc = (const bf16_t*)shortcut; ba.wd = (const bf16_t*)wd; ba.bd = bd;
    const int ptx = (W + 15) / 16, pty = (H + 15) / 16;
    const long np = (long)B * ptx * pty;
    int cus = 256;
    { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
    const long nb = np < cus ? np : cus;
    const size_t lds_bytes = (size_t)9 * 64 * 128 + 2 * 41 * 8 * 128 + 576 * 4;
    hipStream_t s = (hipStream_t)stream;
    if (wd) {
        (void)hipFuncSetAttribute((const void*)bneck_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(bneck_tail_kernel<true>, dim3((unsigned)nb), dim3(512), lds_bytes, s, ba, ptx, pty, (int)np);
    } else {
        (void)hipFuncSetAttribute((const void*)bneck_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(bneck_tail_kernel<false>, dim3((unsigned)nb), dim3(512), lds_bytes, s, ba, ptx, pty, (int)np);
    }
    LOFT_LAUNCH_CHECK();
    return 0;
}

int loft_launch_conv_tap_pipe(const ConvArgs& a, int groups, int mode, int var, int mj, int nw_force, hipStream_t s, int ring32 = 0);     // conv_pipe.hip

struct ConvHead { const void* w; const float* b; float* out; int c4; };

static int conv_tap_bf16_impl(const void* src, const void* wgt, const float* bias, const void* residual,
                              const void* relu_mask, void* out,
                              const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW,
                              int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host,
                              const int* dx_host, const int* wt_host, int relu, int out_f32, int accumulate,
                              int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs,
                              int variant, void* stream, const ConvHead* head) {
    if (T < 1 || T > CONV_MAX_TAPS || (Cin % BK) || (Cout % 4) || groups < 1) return (int)hipErrorInvalidValue;
    const int kern = variant & 0xff;                 // LOFT_CONV_* kernel selector, 0 = the dispatcher's own choice
    if (kern > LOFT_CONV_LEANX || (variant & ~0x3ffff)) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.residual = (const bf16_t*)residual; a.mask = (const bf16_t*)relu_mask;
    a.out = out; a.zero_page = (const bf16_t*)zero_page;
    a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.OHf = OHf; a.OWf = OWf;
    a.os = os; a.oo_y = oo_y; a.oo_x = oo_x; a.ss = ss; a.T = T;
    for (int t = 0; t < T; ++t) { a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t]; }
    a.relu = relu; a.out_f32 = out_f32; a.accumulate = accumulate;
    a.src_gs = src_gs; a.wgt_gs = wgt_gs; a.out_gs = out_gs; a.bias_gs = bias_gs;
    a.trace = nullptr;
    a.nterms = 0; a.amax_x = nullptr; a.amax_w = nullptr; a.amax_out = nullptr;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_c4 = 0; a.par_n = 0;
    if (head) {
        // (served by the staged epilogue of the 256-cout stream tiles only; anything else: the caller launches the head itself)
        if (!head->w || !head->b || !head->out || head->c4 < 4 || head->c4 > 32 || (head->c4 % 4) || Cout != 256 || groups != 1 ||
            out_f32 || accumulate || (variant & ~0xff) || relu_mask)      // (the head reads the tile before a mask would apply)
            return (int)hipErrorInvalidValue;
        a.head_w = (const bf16_t*)head->w; a.head_b = head->b; a.head_out = head->out; a.head_c4 = head->c4;
    }
    fastdiv_setup((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
    fastdiv_setup((unsigned)OW, &a.ow_mul, &a.ow_sh);
    fastdiv_setup((unsigned)B, &a.b_mul, &a.b_sh);
    const long M = (long)B * OH * OW;
    if (M <= 0) return 0;
    if (M > 0x7fffffffL) return (int)hipErrorInvalidValue;
    a.M = (int)M;
    hipStream_t s = (hipStream_t)stream;
    {
        bool p64 = Cin == 64 && Cout == 64 && T >= 4 && T <= 9 && groups == 1 && ss == 1 && os == 1 && OH == IH && OW == IW &&
                   OHf == OH && OWf == OW && oo_y == 0 && oo_x == 0 && !out_f32 && !accumulate;
        for (int t = 0; t < T && p64; ++t) p64 = dy_host[t] >= -1 && dy_host[t] <= 1 && dx_host[t] >= -1 && dx_host[t] <= 1;
        if (kern == LOFT_CONV_PATCH64 && !p64) return (int)hipErrorInvalidValue;
        if (kern == LOFT_CONV_PATCH64 || (kern == LOFT_CONV_AUTO && p64 && M >= 65536)) {
            if (head) return (int)hipErrorInvalidValue;
            Conv64Args c;
            c.src = a.src; c.wgt = a.wgt; c.bias = bias; c.residual = a.residual; c.mask = a.mask; c.out = (bf16_t*)out;
            c.zero_page = a.zero_page; c.B = B; c.H = OH; c.W = OW; c.T = T; c.relu = relu;
            for (int t = 0; t < T; ++t) { c.dy[t] = dy_host[t]; c.dx[t] = dx_host[t]; c.wt[t] = wt_host[t]; }
            const int ptx = (OW + 15) / 16, pty = (OH + 15) / 16;
            const long np = (long)B * ptx * pty;
            int cus = 256;
            { int dev = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
            const long nb = np < cus ? np : cus;
            const size_t lds_bytes = (size_t)T * 64 * 128 + 2 * 41 * 8 * 128;
            hipFuncSetAttribute((const void*)conv64_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(conv64_patch_kernel, dim3((unsigned)nb), dim3(512), lds_bytes, s, c, ptx, pty, (int)np);
            LOFT_LAUNCH_CHECK();
            return 0;
        }
    }
    // Thresholds below are measured on MI355X; `variant` overrides the kernel choice for tests and A/B timing.
    // staged_out: the 128x128 kernels collect a dense bf16 output tile in LDS and store it row-contiguously.
    a.staged_out = !(variant & LOFT_CONV_FLAG_NO_STAGED_OUT) && !accumulate;
    a.pixmajor = 0;
    // Channel-tile-fastest order when the activation is the big operand (it does not fit the 4 MiB L2 of an XCD but the packed
    // weights do): every channel tile of a pixel tile then runs back to back on one XCD and the pixel tile comes from HBM once
    // instead of once per channel tile.
    a.nfast = !(variant & LOFT_CONV_FLAG_NO_NFAST);
    const long big_blocks = (long)loft_cdiv(M, 256) * (Cout / 256) * groups;
    const long Kdim = (long)T * Cin;
    const bool deepk = Kdim >= 1024;      // >= 16 K-steps: hoisted addressing (FAST) amortises its prologue (round 2: 8x64x64 1024->256 32.8 vs 34.6 us)
    // pixel-major enumeration of RoI-map tiles (FAST / pipelined 256-row kernels): taps that leave the map are skipped per tile
    const bool pix_ok = !(variant & LOFT_CONV_FLAG_NO_PIXMAJOR) && T > 1 && T <= 32 && B >= 256 && OH * OW <= 1024 && os == 1 &&
                        ss == 1 && OHf == OH && OWf == OW;
    int k = kern;
    if (k == LOFT_CONV_AUTO) {
        const long half_blocks = (long)loft_cdiv(M, 128) * (Cout / 256) * groups;
        if (Cout % 256 == 0 && big_blocks < 192 && half_blocks >= 192 && Kdim >= 1024 && !out_f32 && !accumulate) {
            // too few 256-pixel tiles to fill the chip (layer3's 64 x 64 maps): the stream kernel on 128-pixel tiles
            k = LOFT_CONV_STREAM128;
        } else if (Cout % 256 == 0 && half_blocks < 192 && (long)loft_cdiv(M, 64) * (Cout / 256) * groups >= 192 && Kdim >= 2048 &&
                   !out_f32 && !accumulate) {
            k = LOFT_CONV_STREAM64;           // layer4's 32 x 32 maps
        } else if (Cout % 256 == 0 && (long)loft_cdiv(M, 64) * (Cout / 256) * groups < 192 &&
                   (long)loft_cdiv(M, 64) * (Cout / 128) * groups >= 192 && Kdim >= 2048 && !out_f32 && !accumulate) {
            k = LOFT_CONV_STREAM64N;          // 256 couts on a 32 x 32 map: 64-pixel x 128-cout tiles fill the chip (P5 3x3: 50 -> 2x as many workgroups)
        } else if (Cout % 256 == 0 && big_blocks >= 192 && !out_f32 && !accumulate) {
            // bf16 output: the software-pipelined kernel with the LDS-staged, row-contiguous epilogue (conv_pipe.hip; +26..43 %
            // over the lockstep 256-tile kernel on the 3x3 / FC shapes, +15..35 % over the 128-tile kernels on the K-shallow 1x1s)
            k = LOFT_CONV_STREAM256;
        } else if (Cout % 256 == 0 && big_blocks >= 192 && Kdim >= 512) {
            k = deepk ? LOFT_CONV_T256_FAST : LOFT_CONV_T256;
        }
        else if (Cout % 128 == 0 && !out_f32 && !accumulate && Kdim >= 1024 && (long)loft_cdiv(M, 256) * (Cout / 128) * groups >= 192) {
            k = LOFT_CONV_STREAM256;          // 128-cout tiles of the stream kernel (layer2's 3x3 convs and their data gradients)
        } else if (Cout % 128 == 0) {
            const bool dense_out = !out_f32 && os == 1 && OHf == OH && OWf == OW;
            const bool two_tiles = residual && relu_mask && dense_out;        // needs both halves of the double buffer
            // (single-stage form at four workgroups per CU: K <= 512 since round 2 -- 8x128x128 512->128: 37.8 against 48.8 us)
            if (!two_tiles && a.staged_out && Kdim <= 512) k = LOFT_CONV_T128_SINGLE;
            else k = deepk ? LOFT_CONV_T128_FAST : LOFT_CONV_T128;
        } else k = LOFT_CONV_T128x64;
    }
    switch (k) {
    case LOFT_CONV_LEANX:
    case LOFT_CONV_LEAN:
    case LOFT_CONV_XFIRST:
    case LOFT_CONV_W4:
    case LOFT_CONV_RING32:
    case LOFT_CONV_ROLES256:
    case LOFT_CONV_STREAM256N:
    case LOFT_CONV_STREAM64N:
    case LOFT_CONV_STREAM64:
    case LOFT_CONV_STREAM128:
    case LOFT_CONV_STREAM256:
    case LOFT_CONV_PIPE256:
        // software-pipelined 256x256 kernels (conv_pipe.hip)
        // (bf16 outputs only: their epilogue collects the output tile in LDS; fp32 / accumulating launches keep the lockstep kernels)
        if ((Cout % 256 && !(k == LOFT_CONV_STREAM256 && Cout % 128 == 0)) || out_f32 || accumulate) return (int)hipErrorInvalidValue;
        if (k == LOFT_CONV_STREAM256N && ((variant >> 12) & 0xf)) return (int)hipErrorInvalidValue;
        if (head && !(k == LOFT_CONV_STREAM256 || k == LOFT_CONV_STREAM128 || k == LOFT_CONV_STREAM64)) return (int)hipErrorInvalidValue;
        a.pixmajor = pix_ok;
        a.pm_S = B; a.pm_P = OH * OW;
        if (pix_ok) {
            // RoI blocks of >= 256 rows per pixel position (ConvArgs::pm_S): padded row count nb * P * S >= P * B
            const int nb = (variant & LOFT_CONV_FLAG_NO_ROI_BLOCKS) ? 1 : std::max(1, B / 256);
            a.pm_S = (B + nb - 1) / nb;
            const long Mp = (long)nb * a.pm_P * a.pm_S;
            if (Mp > 0x7fffffffL) return (int)hipErrorInvalidValue;
            a.M = (int)Mp;
        }
        fastdiv_setup((unsigned)a.pm_S, &a.pms_mul, &a.pms_sh);
        fastdiv_setup((unsigned)a.pm_P, &a.pmp_mul, &a.pmp_sh);
        a.trace = (variant & 0x1000) ? const_cast<float*>(bias) : nullptr;      // experiment bits 12-15 (conv_pipe.hip VAR); TRACE
        if (a.trace) a.bias = nullptr;                                          // borrows the bias pointer for its buffer
        a.tap_major = (variant & LOFT_CONV_FLAG_TAP_MAJOR) ? 1 : 0;
        a.krot = (variant & LOFT_CONV_FLAG_KROT) && !a.tap_major ? 1 : 0;
        if (k == LOFT_CONV_ROLES256 && (Cout % 256 || ((variant >> 12) & 0xf))) return (int)hipErrorInvalidValue;
        if ((k == LOFT_CONV_RING32 || k == LOFT_CONV_W4 || k >= LOFT_CONV_XFIRST) && (Cout % 256 || ((variant >> 12) & 0xf))) return (int)hipErrorInvalidValue;
        return loft_launch_conv_tap_pipe(a, groups, k == LOFT_CONV_PIPE256 ? 0 : (k == LOFT_CONV_ROLES256 ? 2 : 1), (variant >> 12) & 0xf,
                                         k == LOFT_CONV_STREAM128 ? 2 : ((k == LOFT_CONV_STREAM64 || k == LOFT_CONV_STREAM64N) ? 1 : 4),
                                         (k == LOFT_CONV_STREAM64N || k == LOFT_CONV_STREAM256N) ? 1 : 0, s, k == LOFT_CONV_RING32 ? 1 : (k == LOFT_CONV_W4 ? 2 : (k == LOFT_CONV_XFIRST ? 3 : (k == LOFT_CONV_LEAN ? 4 : (k == LOFT_CONV_LEANX ? 5 : 0)))));
    case LOFT_CONV_T256_FAST:
    case LOFT_CONV_T256: {
        if (head) return (int)hipErrorInvalidValue;
        // 256x256 tile, 8 waves of 128x64: half the LDS traffic per FLOP of the 128x128 form; only when it still
        // fills the 256 CUs and K is deep enough (>= 8 K-steps) to amortise the one-block-per-CU prologue/epilogue.
        // (Round-1 measurements of rejected forms -- 4 waves of 128x128, 256x128 tiles with 4 waves, LDS-staged output -- are in
        //  DESIGN.md; note that round 1's pipelining experiments ran with a compiler-inserted vmcnt(0) in front of every
        //  K-step's fragment reads, see conv_tap_kernel.)
        if (Cout % 256) return (int)hipErrorInvalidValue;
        dim3 grid(loft_cdiv(M, 256), Cout / 256, groups);
        if (k == LOFT_CONV_T256_FAST) {
            a.pixmajor = pix_ok;
            hipLaunchKernelGGL((conv_tap_kernel<256, 256, 2, 4, 2, true>), grid, dim3(512), 0, s, a);
        } else hipLaunchKernelGGL((conv_tap_kernel<256, 256, 2, 4>), grid, dim3(512), 0, s, a);
        break;
    }
    case LOFT_CONV_T128_SINGLE:
    case LOFT_CONV_T128_FAST:
    case LOFT_CONV_T128: {
        if (Cout % 128 || head) return (int)hipErrorInvalidValue;
        dim3 grid(loft_cdiv(M, 128), Cout / 128, groups);
        if (k == LOFT_CONV_T128_SINGLE) {
            const bool dense_out = !out_f32 && os == 1 && OHf == OH && OWf == OW;
            if (residual && relu_mask && dense_out) return (int)hipErrorInvalidValue;   // one LDS tile only
            hipLaunchKernelGGL((conv_tap_kernel<128, 128, 2, 2, 1>), grid, dim3(256), 0, s, a);
        } else if (k == LOFT_CONV_T128_FAST)
            hipLaunchKernelGGL((conv_tap_kernel<128, 128, 2, 2, 2, true>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((conv_tap_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, a);
        break;
    }
    case LOFT_CONV_T128x64: {
        if (head) return (int)hipErrorInvalidValue;
        dim3 grid(loft_cdiv(M, 128), loft_cdiv(Cout, 64), groups);
        hipLaunchKernelGGL((conv_tap_kernel<128, 64, 4, 1>), grid, dim3(256), 0, s, a);
        break;
    }
    default:
        return (int)hipErrorInvalidValue;
    }
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_conv_tap_bf16_v(const void* src, const void* wgt, const float* bias, const void* residual,
                                     const void* relu_mask, void* out,
                                     const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW,
                                     int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host,
                                     const int* dx_host, const int* wt_host, int relu, int out_f32, int accumulate,
                                     int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs,
                                     int variant, void* stream) {
    return conv_tap_bf16_impl(src, wgt, bias, residual, relu_mask, out, zero_page, B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo_y,
                              oo_x, ss, T, dy_host, dx_host, wt_host, relu, out_f32, accumulate, groups, src_gs, wgt_gs, out_gs,
                              bias_gs, variant, stream, nullptr);
}

// The same launch with a NARROW 1x1 HEAD on its output computed in the epilogue (ConvArgs block 5): head_out[pixel][n] = head_b[n] +
// sum_c bf16(out[pixel][c]) * head_w[n][c], n < head_c4 (a multiple of 4, <= 32), head_w bf16 [head_c4][256], head_out fp32 [pixels of the full output map][head_c4].  Served by the 256-cout stream tiles only (Cout ==
// 256, one group, bf16 output, the dispatcher's own kernel choice); every other launch returns hipErrorInvalidValue WITHOUT
// launching anything and the caller runs the head as a launch of its own.
LOFT_EXPORT int loft_conv_tap_bf16_head(const void* src, const void* wgt, const float* bias, const void* residual,
                                        const void* relu_mask, void* out,
                                        const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW,
                                        int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host,
                                        const int* dx_host, const int* wt_host, int relu, const void* head_w, const float* head_b,
                                        float* head_out, int head_c4, void* stream) {
    const ConvHead h{head_w, head_b, head_out, head_c4};
    return conv_tap_bf16_impl(src, wgt, bias, residual, relu_mask, out, zero_page, B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo_y,
                              oo_x, ss, T, dy_host, dx_host, wt_host, relu, 0, 0, 1, 0, 0, 0, 0, LOFT_CONV_AUTO, stream, &h);
}

// ConvTranspose2d(kernel 2, stride 2) + bias (+ ReLU) as ONE launch of the stream kernel (fcn_mask_head.py:121-124, the mask head's
// upsampling): out[b, 2y + py, 2x + px, :] = act(W[2 py + px] . x[b, y, x, :] + bias).  The four taps are four N tiles of the same
// pixel tile (ConvArgs::par_n), so x is read from HBM once instead of once per parity launch.  wgt: [4][256][Cin] (tap-major forward
// packing), out: [B, 2H, 2W, 256]; optional narrow head on the output as in loft_conv_tap_bf16_head (head_w == NULL: none).
// Served for Cout == 256 and launches of >= 192 pixel tiles; anything else returns hipErrorInvalidValue WITHOUT launching.
LOFT_EXPORT int loft_deconv2x2_bf16(const void* src, const void* wgt, const float* bias, void* out, const void* zero_page, int B, int H,
                                    int W, int Cin, int Cout, int relu, const void* head_w, const float* head_b, float* head_out,
                                    int head_c4, void* stream) {
    if (Cout != 256 || (Cin % BK) || B < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
    const long M = (long)B * H * W;
    if (M > 0x7fffffffL || loft_cdiv(M, 256) < 192) return (int)hipErrorInvalidValue;
    if (head_w && (!head_b || !head_out || head_c4 < 4 || head_c4 > 32 || (head_c4 % 4))) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.residual = nullptr; a.mask = nullptr;
    a.out = out; a.zero_page = (const bf16_t*)zero_page;
    a.B = B; a.IH = H; a.IW = W; a.Cin = Cin; a.Cout = Cout; a.OH = H; a.OW = W; a.OHf = 2 * H; a.OWf = 2 * W;
    a.os = 2; a.oo_y = 0; a.oo_x = 0; a.ss = 1; a.T = 1;
    a.dy[0] = 0; a.dx[0] = 0; a.wt[0] = 0;
    a.relu = relu; a.out_f32 = 0; a.accumulate = 0;
    a.src_gs = 0; a.wgt_gs = 0; a.out_gs = 0; a.bias_gs = 0;
    a.trace = nullptr;
    a.nterms = 0; a.amax_x = nullptr; a.amax_w = nullptr; a.amax_out = nullptr;
    a.head_w = (const bf16_t*)head_w; a.head_b = head_b; a.head_out = head_out; a.head_c4 = head_w ? head_c4 : 0;
    a.par_n = 1;
    fastdiv_setup((unsigned)(H * W), &a.ohw_mul, &a.ohw_sh);
    fastdiv_setup((unsigned)W, &a.ow_mul, &a.ow_sh);
    fastdiv_setup((unsigned)B, &a.b_mul, &a.b_sh);
    a.M = (int)M;
    a.staged_out = 1; a.pixmajor = 0; a.nfast = 1;
    a.pm_S = B; a.pm_P = H * W;
    fastdiv_setup((unsigned)a.pm_S, &a.pms_mul, &a.pms_sh);
    fastdiv_setup((unsigned)a.pm_P, &a.pmp_mul, &a.pmp_sh);
    a.tap_major = 0; a.krot = 0;
    return loft_launch_conv_tap_pipe(a, 1, 1, 0, 4, 0, (hipStream_t)stream, 0);
}

LOFT_EXPORT int loft_conv_tap_bf16(const void* src, const void* wgt, const float* bias, const void* residual,
                                   const void* relu_mask, void* out,
                                   const void* zero_page, int B, int IH, int IW, int Cin, int Cout, int OH, int OW,
                                   int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T, const int* dy_host,
                                   const int* dx_host, const int* wt_host, int relu, int out_f32, int accumulate,
                                   int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs,
                                   void* stream) {
    return loft_conv_tap_bf16_v(src, wgt, bias, residual, relu_mask, out, zero_page, B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os,
                                oo_y, oo_x, ss, T, dy_host, dx_host, wt_host, relu, out_f32, accumulate, groups, src_gs, wgt_gs,
                                out_gs, bias_gs, LOFT_CONV_AUTO, stream);
}

// The tap convolution on OPERAND PLANES: the fp32 parity mode on the 16-bit matrix cores through the software-pipelined stream
// kernel (conv_pipe.hip, PL instances).  src / wgt hold the 16-bit planes of the fp32 activation / packed fp32 weights
// (loft_split_planes_f32: plane p of src at element offset p * x_ps, of wgt at p * w_ps); term i multiplies activation plane
// xpl[i] with weight plane wpl[i]; all terms of all taps accumulate in fp32 in ONE K loop.  bias, residual, relu_mask, out: fp32,
// semantics of loft_conv_tap_bf16.  amax_x / amax_w: the device scalars the planes were scaled with (both or neither).
// amax_out (optional, PRE-ZEROED): receives max |out| over the elements this launch stores.
// Returns hipErrorInvalidValue for shapes the stream kernel does not serve (Cout % 128, Cin % 64, nterms * T > 64, plane offsets
// beyond 2^31 elements): the caller then takes loft_conv_tap_f32.
LOFT_EXPORT int loft_conv_tap_planes(const void* src, const void* wgt, const float* bias, const float* residual,
                                     const float* relu_mask, float* out, const void* zero_page, int B, int IH, int IW, int Cin,
                                     int Cout, int OH, int OW, int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T,
                                     const int* dy_host, const int* dx_host, const int* wt_host, int relu, int groups,
                                     int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs, int nterms,
                                     const int* xpl_host, const int* wpl_host, int64_t x_ps, int64_t w_ps,
                                     const float* amax_x, const float* amax_w, float* amax_out, void* stream) {
    if (T < 1 || T > CONV_MAX_TAPS || (Cin % BK) || (Cout % 128) || groups < 1 || nterms < 1 || nterms > CONV_MAX_TERMS ||
        nterms * T > 64 || (amax_x == nullptr) != (amax_w == nullptr))
        return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.residual = (const bf16_t*)residual;
    a.mask = (const bf16_t*)relu_mask; a.out = out; a.zero_page = (const bf16_t*)zero_page;
    a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.OHf = OHf; a.OWf = OWf;
    a.os = os; a.oo_y = oo_y; a.oo_x = oo_x; a.ss = ss; a.T = T;
    long max_a = 0, max_w = 0;                    // largest tap offsets (elements): plane offset + tap offset must fit 32 bits
    for (int t = 0; t < T; ++t) {
        a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t];
        max_a = std::max(max_a, labs(((long)dy_host[t] * IW + dx_host[t]) * Cin));
        max_w = std::max(max_w, (long)wt_host[t] * Cout * Cin);
    }
    a.nterms = nterms;
    for (int p = 0; p < nterms; ++p) {
        const long xo = (long)xpl_host[p] * x_ps, wo = (long)wpl_host[p] * w_ps;
        if (xpl_host[p] < 0 || wpl_host[p] < 0 || xo + max_a > 0x7fffffffL || wo + max_w > 0x7fffffffL) return (int)hipErrorInvalidValue;
        a.xoff[p] = (int)xo; a.woff[p] = (int)wo;
    }
    a.amax_x = amax_x; a.amax_w = amax_w; a.amax_out = amax_out;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_c4 = 0; a.par_n = 0;
    a.relu = relu; a.out_f32 = 1; a.accumulate = 0; a.staged_out = 0;
    a.src_gs = src_gs; a.wgt_gs = wgt_gs; a.out_gs = out_gs; a.bias_gs = bias_gs;
    a.trace = nullptr;
    fastdiv_setup((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
    fastdiv_setup((unsigned)OW, &a.ow_mul, &a.ow_sh);
    fastdiv_setup((unsigned)B, &a.b_mul, &a.b_sh);
    const long M = (long)B * OH * OW;
    if (M <= 0) return 0;
    if (M > 0x7fffffffL) return (int)hipErrorInvalidValue;
    a.M = (int)M;
    a.nfast = 1; a.tap_major = 0; a.krot = 0;
    const bool pix_ok = T > 1 && T <= 32 && B >= 256 && OH * OW <= 1024 && os == 1 && ss == 1 && OHf == OH && OWf == OW;
    a.pixmajor = pix_ok;
    a.pm_S = B; a.pm_P = OH * OW;
    if (pix_ok) {
        const int nb = std::max(1, B / 256);
        a.pm_S = (B + nb - 1) / nb;
        const long Mp = (long)nb * a.pm_P * a.pm_S;
        if (Mp > 0x7fffffffL) return (int)hipErrorInvalidValue;
        a.M = (int)Mp;
    }
    fastdiv_setup((unsigned)a.pm_S, &a.pms_mul, &a.pms_sh);
    fastdiv_setup((unsigned)a.pm_P, &a.pmp_mul, &a.pmp_sh);
    // tile shape: the largest of the stream kernel's tiles that still gives ~192 workgroups (as loft_conv_tap_bf16_v)
    int mj, nwf = 0;
    if (Cout % 256 == 0) {
        const long nt = (long)(Cout / 256) * groups;
        if (loft_cdiv(a.M, 256) * nt >= 192) mj = 4;
        else if (loft_cdiv(a.M, 128) * nt >= 192) mj = 2;
        else if (loft_cdiv(a.M, 64) * nt >= 192) mj = 1;
        else { mj = 1; nwf = 1; }
    } else {
        mj = (long)loft_cdiv(a.M, 256) * (Cout / 128) * groups >= 192 ? 4 : 1;
    }
    return loft_launch_conv_tap_pipe(a, groups, 1, 0, mj, nwf, (hipStream_t)stream);
}

// =====================================================================================
// Weight gradient:  dW[wt[t]][n][c] += sum_m  G[b, oy*gos+goy[t], ox*gos+gox[t], n] * X[b, oy*ss+dy[t], ox*ss+dx[t], c]
//
// GEMM view: M' = Cout (n), N' = Cin (c), K' = B*OH*OW pixels -- the reduction runs over pixels, and both
// operands are pixel-major (channels contiguous), i.e. K-strided.  The 64-pixel x 128-channel tiles are
// staged pixel-major with global_load_lds and consumed through ds_read_b64_tr_b16 (gfx950 LDS transpose
// read): a 16-lane group turns a [4 pixel][16 channel] block into "4 consecutive k for my channel",
// which is exactly an MFMA operand quad -- no register shuffles, no transposed copy of the activations.
// Split-K over pixel ranges (grid.z) with fp32 atomic accumulation into dW (caller zeroes it).
// Rows are 256 B = one full bank row, so 16-byte chunk q of pixel row r is stored at q ^ ((r&3)<<2).
// =====================================================================================
// Operand fragment for v_mfma_f32_32x32x16_bf16 from a pixel-major [64][RB/2] bf16 tile (RB bytes per pixel row): lane l ->
// channel col0 + (l&31), pixels kbase + 8*(l>>5) + 0..7, as two transposing reads (4 pixels each).  The reads are INLINE ASM:
// hipcc fences __builtin_amdgcn_ds_read_tr16_b64 behind every outstanding global->LDS copy (`s_waitcnt vmcnt(0)` in front of the
// first fragment read of a K-step, i.e. the prefetch of the next K-step was waited for before computing the current one).  The
// caller waits with TR_WAIT*, naming the registers, before assembling the fragments (cdna_hip_programming.md 5.7).
template <int RB>
__device__ __forceinline__ void tr_frag_issue(const char* tile, int kbase, int col0, int lane, s16x4& lo, s16x4& hi) {
    const int il = lane & 15, gl = lane >> 4;
    const int col = col0 + 16 * (gl & 1) + (il & 3) * 4;
    const int r0 = kbase + 8 * (gl >> 1) + (il >> 2);
    const int r1 = r0 + 4;
    const unsigned p0 = (unsigned)(size_t)(tile + r0 * RB + wswz(r0, col >> 3) * 16 + (col & 7) * 2);
    const unsigned p1 = (unsigned)(size_t)(tile + r1 * RB + wswz(r1, col >> 3) * 16 + (col & 7) * 2);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(p0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(p1));
}
__device__ __forceinline__ bf16x8 tr_join(const s16x4 lo, const s16x4 hi) {
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
#define TR_WAIT4(l, h) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((l)[0]), "+v"((l)[1]), "+v"((l)[2]), "+v"((l)[3]), \
                                    "+v"((h)[0]), "+v"((h)[1]), "+v"((h)[2]), "+v"((h)[3]))
#define TR_WAIT6(l, h) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((l)[0]), "+v"((l)[1]), "+v"((l)[2]), "+v"((l)[3]), "+v"((l)[4]), \
                                    "+v"((l)[5]), "+v"((h)[0]), "+v"((h)[1]), "+v"((h)[2]), "+v"((h)[3]), "+v"((h)[4]), "+v"((h)[5]))

// TN = output tile edge (n and c), NW = waves.  <128,4>: 2x2 waves of 64x64; <256,8>: 2x4 waves of 128(n)x64(c) --
// the 256 form halves the LDS-DMA bytes per FLOP and is used when Cout and Cin are multiples of 256.
template <int TN, int NW, bool PM = false>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int RB = TN * 2;                 // bytes per pixel row of a tile
    constexpr int CPR = RB / 16;               // 16-byte chunks per row
    constexpr int RPW = 1024 / RB;             // tile rows covered by one wave-level glds
    constexpr int TILE_BYTES = 64 * RB;
    constexpr int WAVES_C = NW / 2;
    constexpr int NI = TN / 2 / 32, NJ = TN / WAVES_C / 32;
    static_assert(NJ == 2 && WAVES_C >= NI, "tiling");
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [buf][G|X]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bx = V % gridDim.x, by = (V / gridDim.x) % gridDim.y, bz = V / (gridDim.x * gridDim.y);
    const int nt = bx / a.ctiles, ct = bx - nt * a.ctiles;
    int t, grp, mbeg, mend;
    if constexpr (PM) {
        grp = by;
        t = a.pm_tap[bz];
        mbeg = 0;
        mend = a.pm_rows[t];            // local rows of this split: (position, RoI of the split's range); see WgradArgs
    } else {
        t = by % a.T; grp = by / a.T;
        mbeg = bz * a.pix_per_split;
        mend = min(a.M, mbeg + a.pix_per_split);
    }
    const int n0 = nt * TN, c0 = ct * TN;
    if (mbeg >= mend) return;
    const bf16_t* G = a.g + WGRAD_G_OFF(a, grp);
    const bf16_t* X = a.x + WGRAD_X_OFF(a, grp);
    const int ohw = a.OH * a.OW;
    const int goy = a.goy[t], gox = a.gox[t], dy = a.dy[t], dx = a.dx[t];
    const int pm_y0 = PM ? a.pm_y0[t] : 0, pm_x0 = PM ? a.pm_x0[t] : 0, pm_rw = PM ? a.pm_rw[t] : 1;
    const unsigned pm_mul = PM ? a.pm_rw_mul[t] : 0u, pm_sh = PM ? a.pm_rw_sh[t] : 0u;
    const int pm_rb = PM ? a.pm_pps[t] : 1, pm_b0 = PM ? (int)a.pm_split[bz] * pm_rb : 0;
    const unsigned rb_mul = PM ? a.pm_pps_mul[t] : 0u, rb_sh = PM ? a.pm_pps_sh[t] : 0u;

    const int lrow = lane / CPR, lchunk = lane % CPR;
    auto stage = [&](int m_base, int buf) {
        char* gbuf = lds + buf * 2 * TILE_BYTES;
        char* xbuf = gbuf + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + wave * RPW + lrow;
            const int m = m_base + row;
            const int q = wswz(row, lchunk) * 8;
            const bf16_t* pg = a.zero_page;
            const bf16_t* px = a.zero_page;
            if (m < mend) {
                if constexpr (PM) {     // m = (position inside the valid rectangle of tap t) * RoIs-per-split + RoI; stride 1, in bounds
                    const int p = fastdiv(m, rb_mul, rb_sh), b = pm_b0 + (m - p * pm_rb);
                    const int ry = fastdiv(p, pm_mul, pm_sh), rx = p - ry * pm_rw;
                    const int oy = pm_y0 + ry, ox = pm_x0 + rx;
                    if (b < a.B) {
                        pg = G + ((long)(b * a.GH + oy + goy) * a.GW + ox + gox) * a.Cout + n0 + q;
                        px = X + ((long)(b * a.XH + oy + dy) * a.XW + ox + dx) * a.Cin + c0 + q;
                    }
                } else {
                const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                const int gy = oy * a.gos + goy, gx = ox * a.gos + gox;
                const int iy = oy * a.ss + dy, ix = ox * a.ss + dx;
                if ((gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                    pg = G + ((long)(b * a.GH + gy) * a.GW + gx) * a.Cout + n0 + q;
                    px = X + ((long)(b * a.XH + iy) * a.XW + ix) * a.Cin + c0 + q;
                }
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)pg, (lds_ptr_t)(gbuf + (i * 16 + wave * RPW) * RB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)px, (lds_ptr_t)(xbuf + (i * 16 + wave * RPW) * RB), 16, 0, 0);
        }
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wn = wave / WAVES_C, wc = wave % WAVES_C;
    // bias gradient rides along on the blocks of the first channel tile: one extra MFMA per K sub-step against an
    // all-ones operand gives sum_k G[k][n] in every column of the result; wave (wn, wc) takes n-tile wn*NI + wc.
    const bool do_db = WGRAD_DB_ON(a, grp) && ct == 0 && wc < NI && (a.db_tap == -2 || a.db_tap == t);
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)LOFT_ONE16;
    const int nsteps = (mend - mbeg + 63) / 64;
    stage(mbeg, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) stage(mbeg + (s + 1) * 64, (s + 1) & 1);
        const char* gbuf = lds + (s & 1) * 2 * TILE_BYTES;
        const char* xbuf = gbuf + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 gf[NI], xf[NJ];
            s16x4 flo[NI + NJ], fhi[NI + NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) tr_frag_issue<RB>(gbuf, ks * 16, wn * (TN / 2) + i * 32, lane, flo[i], fhi[i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) tr_frag_issue<RB>(xbuf, ks * 16, wc * (TN / WAVES_C) + j * 32, lane, flo[NI + j], fhi[NI + j]);
            if constexpr (NI + NJ == 4) TR_WAIT4(flo, fhi); else TR_WAIT6(flo, fhi);
#pragma unroll
            for (int i = 0; i < NI; ++i) gf[i] = tr_join(flo[i], fhi[i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) xf[j] = tr_join(flo[NI + j], fhi[NI + j]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = LOFT_MFMA_32x32x16(gf[i], xf[j], acc[i][j]);
            if (do_db) {   // static register selects (a runtime-indexed fragment array would be demoted to scratch)
                bf16x8 gsel = gf[0];
#pragma unroll
                for (int i = 1; i < NI; ++i) gsel = (wc == i) ? gf[i] : gsel;
                accb = LOFT_MFMA_32x32x16(gsel, ones, accb);
            }
        }
    }

    if (do_db && (lane & 31) == 0) {
        float* db = WGRAD_DB_PTR(a, grp);
        const float dbs = WGRAD_DB_SCALE(a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * (TN / 2) + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            unsafeAtomicAdd(db + n, accb[r] * dbs);
        }
    }

    const int sidx = PM ? (int)a.pm_split[bz] : bz;
    const int ns_t = PM ? a.pm_blk0[t + 1] - a.pm_blk0[t] : (int)gridDim.z;
    float* dw = a.dw + WGRAD_DW_OFF(a, grp) + (long)a.wt[t] * a.Cout * a.Cin + (a.partial ? (long)sidx * a.split_stride : 0l);
    const int nzero = (a.partial && sidx == ns_t - 1) ? a.nslots - ns_t : 0;      // this tap's unused slots (WgradArgs::partial)
    const float osc = WGRAD_OUT_SCALE(a);          // (1 unless the operands are scaled planes: exact either way)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = c0 + wc * (TN / WAVES_C) + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (TN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float* p = dw + (long)n * a.Cin + c;
                if (a.partial) {
                    *p = acc[i][j][r];
                    for (int z = 1; z <= nzero; ++z) p[(long)z * a.split_stride] = 0.f;
                } else {
                    unsafeAtomicAdd(p, acc[i][j][r] * osc);
                }
            }
        }
}


// Narrow-channel form (Cout or Cin not a multiple of 128: HRNet branches of 32/64 channels, its 64-channel stem).  One 256-byte
// LDS row per pixel holds BOTH operands side by side -- columns 0..63 = 64 channels of G, columns 64..127 = 64 channels of X --
// so the same bank-conflict-free swizzle and the same transposing fragment reader apply.  2 x 2 waves of one 32 x 32 MFMA tile;
// channels past Cout / Cin read the zero page and are masked in the epilogue.
// MODE 1 (dense: one zero-offset tap, unit strides, one map size) / 2 (same: stride-1 same-size taps, OW >= 64): the decode-free
// row addressing of conv_wgrad_ring_kernel (conv_wgrad_pipe.hip) -- reduction row m is pixel m of G, pixel m (+ dy * W + dx) of X.
template <int MODE>
__global__ __launch_bounds__(256) void conv_wgrad64_kernel(const WgradArgs a) {
    constexpr bool DENSE = MODE == 1, SAME = MODE == 2;
    constexpr int RB = 256, TILE_BYTES = 64 * RB;
    __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bx = V % gridDim.x, by = (V / gridDim.x) % gridDim.y, bz = V / (gridDim.x * gridDim.y);
    const int nt = bx / a.ctiles, ct = bx - nt * a.ctiles;
    const int t = by % a.T, grp = by / a.T;
    const int n0 = nt * 64, c0 = ct * 64;
    const int mbeg = bz * a.pix_per_split;
    const int mend = min(a.M, mbeg + a.pix_per_split);
    if (mbeg >= mend) return;
    const bf16_t* G = a.g + WGRAD_G_OFF(a, grp);
    const bf16_t* X = a.x + WGRAD_X_OFF(a, grp);
    const int ohw = a.OH * a.OW;
    const int goy = a.goy[t], gox = a.gox[t], dy = a.dy[t], dx = a.dx[t];
    const int lrow = lane >> 4, lchunk = lane & 15;
    int s_oy[4] = {0, 0, 0, 0}, s_ox[4] = {0, 0, 0, 0};       // SAME: map position of this thread's rows at the next stage() call
    const int xshift = dy * a.XW + dx;
    if constexpr (SAME) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbeg + i * 16 + wave * 4 + lrow;
            const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
            s_oy[i] = fastdiv(rem, a.ow_mul, a.ow_sh);
            s_ox[i] = rem - s_oy[i] * a.OW;
        }
    }
    auto stage = [&](int m_base, int buf) {
        char* tb = lds + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + wave * 4 + lrow;
            const int m = m_base + row;
            const int q = wswz(row, lchunk);                    // logical 16-byte chunk held at this physical position
            const bf16_t* p = a.zero_page;
            if constexpr (DENSE || SAME) {
                bool ok = m < mend;
                if constexpr (SAME) {
                    const int iy = s_oy[i] + dy, ix = s_ox[i] + dx;
                    ok = ok & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW);
                    s_ox[i] += 64;
                    if (s_ox[i] >= a.OW) {
                        s_ox[i] -= a.OW;
                        s_oy[i] = s_oy[i] + 1 == a.OH ? 0 : s_oy[i] + 1;
                    }
                }
                if (ok) {
                    if (q < 8) {
                        if (n0 + q * 8 < a.Cout) p = G + (long)m * a.Cout + n0 + q * 8;
                    } else {
                        if (c0 + (q - 8) * 8 < a.Cin) p = X + (long)(m + (SAME ? xshift : 0)) * a.Cin + c0 + (q - 8) * 8;
                    }
                }
            } else
            if (m < mend) {
                const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                const int gy = oy * a.gos + goy, gx = ox * a.gos + gox;
                const int iy = oy * a.ss + dy, ix = ox * a.ss + dx;
                if ((gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                    if (q < 8) {
                        if (n0 + q * 8 < a.Cout) p = G + ((long)(b * a.GH + gy) * a.GW + gx) * a.Cout + n0 + q * 8;
                    } else {
                        if (c0 + (q - 8) * 8 < a.Cin) p = X + ((long)(b * a.XH + iy) * a.XW + ix) * a.Cin + c0 + (q - 8) * 8;
                    }
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(tb + (i * 16 + wave * 4) * RB), 16, 0, 0);
        }
    };
    f32x16 acc, accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
    const int wn = wave >> 1, wc = wave & 1;
    const bool do_db = WGRAD_DB_ON(a, grp) && ct == 0 && wc == 0 && (a.db_tap == -2 || a.db_tap == t);
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)LOFT_ONE16;
    const int nsteps = (mend - mbeg + 63) / 64;
    stage(mbeg, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) stage(mbeg + (s + 1) * 64, (s + 1) & 1);
        const char* tb = lds + (s & 1) * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s16x4 flo[2], fhi[2];
            tr_frag_issue<RB>(tb, ks * 16, wn * 32, lane, flo[0], fhi[0]);
            tr_frag_issue<RB>(tb, ks * 16, 64 + wc * 32, lane, flo[1], fhi[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(flo[0]), "+v"(flo[1]), "+v"(fhi[0]), "+v"(fhi[1]));
            const bf16x8 gf = tr_join(flo[0], fhi[0]);
            const bf16x8 xf = tr_join(flo[1], fhi[1]);
            acc = LOFT_MFMA_32x32x16(gf, xf, acc);
            if (do_db) accb = LOFT_MFMA_32x32x16(gf, ones, accb);
        }
    }
    if (do_db && (lane & 31) == 0) {
        float* db = WGRAD_DB_PTR(a, grp);
        const float dbs = WGRAD_DB_SCALE(a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < a.Cout) unsafeAtomicAdd(db + n, accb[r] * dbs);
        }
    }
    float* dw = a.dw + WGRAD_DW_OFF(a, grp) + (long)a.wt[t] * a.Cout * a.Cin;
    const int c = c0 + wc * 32 + (lane & 31);
    if (c < a.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < a.Cout) unsafeAtomicAdd(dw + (long)n * a.Cin + c, acc[r]);
        }
    }
}

// All-taps form for the narrow stride-1 same-size convs at high resolution (HRNet's 32/64-channel 3x3 branches at 256^2 / 128^2):
// the tap-parallel kernels above give every (tap, pixel range) its own workgroup, so with a 64 x 64 output tile a K-step is
// 16 KB of copies for 4 MFMAs per wave -- the loop runs at one global->LDS round trip per 64 pixels (228 us for 8 x 256^2 pixels,
// 0.6 TB/s).  Here a workgroup walks 8x8-PIXEL PATCHES: it stages the patch of G (64 rows) and the 10x10 halo of X (100 rows)
// once and feeds ALL taps from them -- the MFMA k index runs over the patch's pixels, tap (dy, dx) just reads X rows shifted
// inside the halo (the transposing reader takes one address per lane, so any row map is free): 21 KB of copies for 36 MFMAs.
// 128-byte LDS rows, 16-byte chunk q of row r stored at q ^ (((r >> 1) & 1) << 2) (rows r and r+2 would share banks).
__device__ __forceinline__ int psw(int row) { return ((row >> 1) & 1) << 2; }

__device__ __forceinline__ bf16x8 patch_frag(const char* tile, int r0, int r1, int col) {
    // lane's 4 columns col..col+3 (8 bytes) of LDS rows r0 / r1 -> "8 consecutive k for my column" after the transpose
    const char* p0 = tile + r0 * 128 + (((col >> 3) ^ psw(r0)) << 4) + (col & 7) * 2;
    const char* p1 = tile + r1 * 128 + (((col >> 3) ^ psw(r1)) << 4) + (col & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))      // 160 accumulator registers + <= 96 others: two workgroups per CU
void conv_wgrad64_patch_kernel(const WgradArgs a, int ptx, int pty, int npatch,
                                                                 float* __restrict__ partial) {
    constexpr int GB = 64 * 128, XB = 104 * 128, STG = GB + XB;
    __shared__ __attribute__((aligned(16))) char lds[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.y;
    const bf16_t* G = a.g + WGRAD_G_OFF(a, grp);
    const bf16_t* X = a.x + WGRAD_X_OFF(a, grp);
    const int H = a.OH, W = a.OW;
    const int lrow = lane >> 3, pc = lane & 7;
    auto stage = [&](int p, int buf) {
        char* gb = lds + buf * STG;
        char* xb = gb + GB;
        const int b = p / (ptx * pty), rem = p - b * (ptx * pty);
        const int y0 = (rem / ptx) * 8, x0 = (rem % ptx) * 8;
#pragma unroll
        for (int it = 0; it < 2; ++it) {                       // G: 64 patch pixels, 8 rows per wave-level copy
            const int row = it * 32 + wave * 8 + lrow;
            const int q = pc ^ psw(row);
            const int y = y0 + (row >> 3), x = x0 + (row & 7);
            const bf16_t* ptr = a.zero_page;
            if (y < H && x < W && q * 8 < a.Cout) ptr = G + ((long)(b * H + y) * W + x) * a.Cout + q * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lds_ptr_t)(gb + (it * 32 + wave * 8) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {                       // X: 10 x 10 halo (13 copies of 8 rows; rows >= 100 unused)
            const int slot = it * 4 + wave;
            if (slot < 13) {
                const int row = slot * 8 + lrow;
                const int q = pc ^ psw(row);
                const int hy = row / 10, hx = row - hy * 10;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bf16_t* ptr = a.zero_page;
                if (row < 100 && y >= 0 && y < H && x >= 0 && x < W && q * 8 < a.Cin) ptr = X + ((long)(b * H + y) * W + x) * a.Cin + q * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)ptr, (lds_ptr_t)(xb + slot * 8 * 128), 16, 0, 0);
            }
        }
    };
    f32x16 acc[9], accb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const int wn = wave >> 1, wc = wave & 1;
    const bool do_db = a.db != nullptr && wc == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)LOFT_ONE16;
    // this lane's fragment rows inside a 16-pixel k block: patch row 2*ks + (gl >> 1), patch columns (il >> 2) and (il >> 2) + 4
    const int il = lane & 15, gl = lane >> 4;
    const int fcol = 16 * (gl & 1) + (il & 3) * 4;
    const int fpy = gl >> 1, fpx = il >> 2;
    int p = blockIdx.x;
    if (p < npatch) stage(p, 0);
    for (int s = 0; p < npatch; p += gridDim.x, ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p + (int)gridDim.x < npatch) stage(p + gridDim.x, (s + 1) & 1);
        const char* gb = lds + (s & 1) * STG;
        const char* xb = gb + GB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int py = 2 * ks + fpy;                       // k = py * 8 + px
            const int gr0 = py * 8 + fpx;
            const bf16x8 gf = patch_frag(gb, gr0, gr0 + 4, wn * 32 + fcol);
            if (do_db) accb = LOFT_MFMA_32x32x16(gf, ones, accb);
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (t < a.T) {
                    const int hr0 = (py + 1 + a.dy[t]) * 10 + fpx + 1 + a.dx[t];
                    const bf16x8 xf = patch_frag(xb, hr0, hr0 + 4, wc * 32 + fcol);
                    acc[t] = LOFT_MFMA_32x32x16(gf, xf, acc[t]);
                }
        }
    }
    // per-workgroup partial sums [group][block][T * Cout * Cin + Cout] (plain stores; 36 864 atomics per workgroup cost more than
    // the kernel itself: 1024 workgroups +24 ms per HRNet step), summed by conv_wgrad_patch_reduce_kernel
    const long pstride = (long)a.T * a.Cout * a.Cin + a.Cout;
    float* pp = partial + ((long)grp * gridDim.x + blockIdx.x) * pstride;
    if (do_db && (lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < a.Cout) pp[(long)a.T * a.Cout * a.Cin + n] = accb[r];
        }
    }
    const int c = wc * 32 + (lane & 31);
    if (c < a.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (t < a.T) {
                float* dw = pp + (long)a.wt[t] * a.Cout * a.Cin;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n < a.Cout) dw[(long)n * a.Cin + c] = acc[t][r];
                }
            }
    }
}

// dw[g][i] += sum_b partial[g][b][i], db[g][n] += sum_b partial[g][b][T*Cout*Cin + n]
__global__ void conv_wgrad_patch_reduce_kernel(const float* __restrict__ partial, int nb, long pstride, long nw, int Cout,
                                               float* __restrict__ dw, long dw_gs, float* __restrict__ db) {
    // grid.z slices of 16 partials each: 16 loads per thread, then one atomic (a single-pass sum over 512 partials per thread is
    // a 512-deep chain of dependent-latency loads)
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (i >= pstride) return;
    const int b0 = blockIdx.z * 16, b1 = min(nb, b0 + 16);
    const float* p = partial + (long)g * nb * pstride + i;
    float sacc = 0.f;
    for (int b = b0; b < b1; ++b) sacc += p[(long)b * pstride];
    if (i < nw) unsafeAtomicAdd(dw + (long)g * dw_gs + i, sacc);
    else if (db) unsafeAtomicAdd(db + (long)g * Cout + (i - nw), sacc);
}

LOFT_EXPORT int64_t loft_conv_wgrad_patch_workspace_bytes(int B, int H, int W, int Cout, int Cin, int T, int groups) {
    const long np = (long)B * ((W + 7) / 8) * ((H + 7) / 8);
    const long nb = np < 512 ? np : 512;
    return (int64_t)groups * nb * ((long)T * Cout * Cin + Cout) * 4;
}

// Weight (+ bias) gradient of a stride-1, same-size conv with <= 64 input and output channels and taps within +-1 pixel
// (see conv_wgrad64_patch_kernel).  dw [groups][T][Cout][Cin] / db [groups][Cout] are ACCUMULATED into.
LOFT_EXPORT int loft_conv_wgrad_patch_bf16(const void* g, const void* x, float* dw, const void* zero_page, int B, int H, int W,
                                           int Cout, int Cin, int T, const int* dy_host, const int* dx_host, const int* wt_host,
                                           int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs, float* db, void* workspace,
                                           void* stream) {
    if (T < 1 || T > 9 || Cout > 64 || Cin > 64 || (Cin % 8) || (Cout % 8) || groups < 1 || !workspace) return (int)hipErrorInvalidValue;
    WgradArgs a;
    a.pm_inc_ok = 0;
    a.nvg = 0; a.amax_g = nullptr; a.amax_x = nullptr;
    a.g = (const bf16_t*)g; a.x = (const bf16_t*)x; a.dw = dw; a.zero_page = (const bf16_t*)zero_page;
    a.B = B; a.GH = H; a.GW = W; a.Cout = Cout; a.XH = H; a.XW = W; a.Cin = Cin; a.OH = H; a.OW = W;
    a.gos = 1; a.ss = 1; a.T = T;
    for (int t = 0; t < T; ++t) {
        if (dy_host[t] < -1 || dy_host[t] > 1 || dx_host[t] < -1 || dx_host[t] > 1) return (int)hipErrorInvalidValue;
        a.goy[t] = 0; a.gox[t] = 0; a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t];
    }
    a.g_gs = g_gs; a.x_gs = x_gs; a.dw_gs = dw_gs;
    a.M = B * H * W; a.pix_per_split = 0; a.ctiles = 1;
    a.ohw_mul = a.ohw_sh = a.ow_mul = a.ow_sh = 0;
    a.db = db; a.db_tap = -2;
    const int ptx = (W + 7) / 8, pty = (H + 7) / 8;
    const long np = (long)B * ptx * pty;
    const long nb = np < 512 ? np : 512;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad64_patch_kernel, dim3((unsigned)nb, groups), dim3(256), 0, s, a, ptx, pty, (int)np, (float*)workspace);
    LOFT_LAUNCH_CHECK();
    const long nw = (long)T * Cout * Cin, pstride = nw + Cout;
    hipLaunchKernelGGL(conv_wgrad_patch_reduce_kernel, dim3(loft_cdiv(pstride, 256), groups, loft_cdiv(nb, 16)), dim3(256), 0, s, (const float*)workspace,
                       (int)nb, pstride, nw, Cout, dw, (long)dw_gs, db);
    LOFT_LAUNCH_CHECK();
    return 0;
}


int loft_launch_conv_wgrad_stream(const WgradArgs& a, dim3 grid, bool pm, hipStream_t s);     // conv_wgrad_pipe.hip
int loft_launch_conv_wgrad_ring(const WgradArgs& a, dim3 grid, hipStream_t s);                 // conv_wgrad_pipe.hip

// mode 0: launch, split-K combined with fp32 atomics into the zeroed dw.  mode 1: no launch, *nslots_out = the number of
// split slots a partial-sum launch of this shape writes (0: this shape has no such form -- narrow channels, repeated or missing
// weight taps, a tap without a valid row).  mode 2: launch, every workgroup STORES its tile into its split's slot of
// dw = [group][nslots][T][Cout][Cin] (dw_gs = nslots * T * Cout * Cin); see WgradArgs::partial.
// Operand planes of a weight-gradient launch (loft_conv_wgrad_planes): the launch's groups are `groups * nterms` virtual groups,
// virtual group grp * nterms + p reads G plane gpl[p] / X plane xpl[p] of real group grp and adds into that group's dW.
struct WgradPlanes {
    int nterms;
    const int* gpl;
    const int* xpl;
    int64_t g_ps, x_ps;              // elements between two planes of G / X
    const float* amax_g;
    const float* amax_x;
};

static int wgrad_impl(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH,
                      int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                      const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                      const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                      int splits, float* db, int db_tap, int variant, void* stream, int mode, int* nslots_out,
                      const WgradPlanes* planes = nullptr) {
    if (T < 1 || T > CONV_MAX_TAPS || (Cin % 8) || (Cout % 8) || groups < 1) return (int)hipErrorInvalidValue;
    if (variant < LOFT_WGRAD_AUTO || variant > LOFT_WGRAD_RING128) return (int)hipErrorInvalidValue;
    const bool narrow = (Cin % 128) || (Cout % 128);
    bool slots_ok = !narrow;
    {   // every weight tap written by exactly one tap of the table (else a slot would be written twice, or never)
        unsigned seen = 0;
        for (int t = 0; t < T; ++t) {
            if (wt_host[t] < 0 || wt_host[t] >= T || (seen >> wt_host[t] & 1u)) slots_ok = false;
            else seen |= 1u << wt_host[t];
        }
    }
    if (nslots_out) *nslots_out = 0;
    if (mode == 2 && !slots_ok) return (int)hipErrorInvalidValue;
    WgradArgs a;
    a.pm_inc_ok = 0;
    a.nvg = 0; a.amax_g = nullptr; a.amax_x = nullptr;
    if (planes) {
        if (mode != 0 || planes->nterms < 1 || planes->nterms > CONV_MAX_TERMS || groups * planes->nterms > WGRAD_MAX_VGROUPS)
            return (int)hipErrorInvalidValue;
        if (db) {
            // the bias gradient of a plane launch = the column sums of ALL planes of G: every G plane that occurs in a term must be
            // paired with X plane 0 in exactly one term -- that term's workgroups carry its sum (true of the three term lists)
            unsigned seen = 0u, want = 0u;
            for (int p = 0; p < planes->nterms; ++p) {
                if (planes->gpl[p] < 0 || planes->gpl[p] > 31) return (int)hipErrorInvalidValue;
                want |= 1u << planes->gpl[p];
                if (planes->xpl[p] == 0) {
                    if (seen >> planes->gpl[p] & 1u) return (int)hipErrorInvalidValue;
                    seen |= 1u << planes->gpl[p];
                }
            }
            if (seen != want) return (int)hipErrorInvalidValue;
        }
        a.nvg = groups * planes->nterms;
        for (int gr = 0; gr < groups; ++gr)
            for (int p = 0; p < planes->nterms; ++p) {
                const int v = gr * planes->nterms + p;
                a.vg_g[v] = (long)gr * g_gs + (long)planes->gpl[p] * planes->g_ps;
                a.vg_x[v] = (long)gr * x_gs + (long)planes->xpl[p] * planes->x_ps;
                a.vg_dw[v] = (long)gr * dw_gs;
                a.vg_db[v] = (db && planes->xpl[p] == 0) ? gr * Cout : -1;
            }
        a.amax_g = planes->amax_g; a.amax_x = planes->amax_x;
        groups = a.nvg;                       // from here on: launch geometry and split counts over the virtual groups
    }
    a.partial = mode == 2; a.nslots = 1; a.split_stride = (long)T * Cout * Cin;
    a.g = (const bf16_t*)g; a.x = (const bf16_t*)x; a.dw = dw; a.zero_page = (const bf16_t*)zero_page;
    a.B = B; a.GH = GH; a.GW = GW; a.Cout = Cout; a.XH = XH; a.XW = XW; a.Cin = Cin; a.OH = OH; a.OW = OW;
    a.gos = gos; a.ss = ss; a.T = T;
    for (int t = 0; t < T; ++t) {
        a.goy[t] = goy_host[t]; a.gox[t] = gox_host[t]; a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t];
    }
    a.g_gs = g_gs; a.x_gs = x_gs; a.dw_gs = dw_gs;
    a.db = db; a.db_tap = db ? db_tap : -1;
    const long M = (long)B * OH * OW;
    if (M <= 0) return mode == 2 ? (int)hipErrorInvalidValue : 0;
    if (M > 0x7fffffffL) return (int)hipErrorInvalidValue;
    a.M = (int)M;
    fastdiv_setup((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
    fastdiv_setup((unsigned)OW, &a.ow_mul, &a.ow_sh);
    // 256x256 tiles only when >= 256 workgroups can each run >= ~32 K-steps (else the 65k-atomic epilogue dominates)
    const bool big_ok = (Cout % 256 == 0) && (Cin % 256 == 0);
    if ((variant == LOFT_WGRAD_STREAM256 || variant == LOFT_WGRAD_T256) && !big_ok) return (int)hipErrorInvalidValue;
    if ((variant == LOFT_WGRAD_T128 || variant == LOFT_WGRAD_RING128) && narrow) return (int)hipErrorInvalidValue;
    const bool big = variant == LOFT_WGRAD_AUTO ? (big_ok && M * (long)(Cout / 256) * (Cin / 256) * T * groups >= 524288L)
                                                : (variant != LOFT_WGRAD_T128 && variant != LOFT_WGRAD_RING128);
    const bool piped = big && variant != LOFT_WGRAD_T256;       // the software-pipelined form (conv_wgrad_pipe.hip)
    const int TNv = narrow ? 64 : (big ? 256 : 128);
    a.ctiles = (Cin + TNv - 1) / TNv;
    const int tiles = ((Cout + TNv - 1) / TNv) * a.ctiles;
    if (splits <= 0) {
        // split-K factor: ~512 workgroups (two resident 128-tile workgroups per CU).  More splits only add fp32 atomic
        // traffic to dW -- measured on the 1x1 shapes: 1024 workgroups 202-222 TFLOP/s, 512: 279-305, 256: 261-274.
        // (the four-stage ring kernel with its decode-free row addressing: ~288 workgroups are best on every backbone shape,
        //  tools/probes/wgrad_splits.py -- layer3 1x1 37.6 us at 256 against 46 at 512, layer3 3x3 78 at 288 against 91 / 99)
        const bool ring = !narrow && variant != LOFT_WGRAD_T128 && gos == 1 && ss == 1 && GH == OH && GW == OW && XH == OH && XW == OW;
        const long tgt_small = ring ? 288 : 512;
        const long tgt_big = 256;   // 256-tile: one workgroup per CU
        // (big tile, measured: 256 workgroups 708-791 TFLOP/s on the FOA / mask / P2-P3 3x3 shapes, 512: 606-754, 1024: 474-719)
        // (also measured: a single-stage 128-tile form at four workgroups per CU -- what helped the K-shallow forward convs --
        //  changes nothing here, with 512 or 1024 workgroups: these launches are bound by the fp32-atomic epilogue and the
        //  operand re-reads, not by the per-K-step round trip)
        long want = (big ? tgt_big : tgt_small) / ((long)tiles * T * groups);
        long maxs = (M + 255) / 256;
        splits = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
    }
    int pps = (int)((M + splits - 1) / splits);
    pps = ((pps + 63) / 64) * 64;
    a.pix_per_split = pps;
    if (!narrow && T > 1 && gos == 1 && ss == 1 && B >= 128 && OH * OW <= 1024) {
        // valid rectangle of every tap; K-splits over RoI ranges (WgradArgs: split s of tap t = RoIs [s * rb_t, (s + 1) * rb_t) at all
        // of the tap's positions), rb_t chosen per tap for a common K length: the smallest L (rows per split, multiple of the
        // 64-row K-step) with sum_t ceil(B / floor(L / positions_t)) <= the workgroup budget per group
        const long budget = std::min<long>((long)splits * T, WGRAD_PM_MAX_BLOCKS);    // (pm_tap / pm_split hold one byte per block)
        long npos[CONV_MAX_TAPS], total = 0;
        for (int t = 0; t < T; ++t) {
            const int ylo = std::max(0, std::max(-a.goy[t], -a.dy[t])), yhi = std::min(OH, std::min(GH - a.goy[t], XH - a.dy[t]));
            const int xlo = std::max(0, std::max(-a.gox[t], -a.dx[t])), xhi = std::min(OW, std::min(GW - a.gox[t], XW - a.dx[t]));
            const int rh = std::max(0, yhi - ylo), rw = std::max(0, xhi - xlo);
            a.pm_y0[t] = ylo; a.pm_x0[t] = xlo; a.pm_rw[t] = rw > 0 ? rw : 1;
            fastdiv_setup((unsigned)a.pm_rw[t], &a.pm_rw_mul[t], &a.pm_rw_sh[t]);
            npos[t] = (long)rh * rw;
            total += npos[t] * B;
        }
        if (total > 0) {
            long L = ((total + budget - 1) / budget + 63) / 64 * 64;
            for (;; L += 64) {
                long nb = 0;
                for (int t = 0; t < T; ++t)
                    if (npos[t] > 0) {
                        const long rb = std::max<long>(1, L / npos[t]);
                        nb += (B + rb - 1) / rb;
                    }
                if (nb <= budget) break;
            }
            int blk = 0;
            for (int t = 0; t < T; ++t) {
                long ns = 0, rb = 1;
                if (npos[t] > 0) {
                    rb = std::max<long>(1, L / npos[t]);
                    ns = (B + rb - 1) / rb;
                    rb = (B + ns - 1) / ns;                      // same number of splits, evenly sized RoI ranges
                }
                a.pm_blk0[t] = blk;
                a.pm_pps[t] = (int)rb;
                a.pm_rows[t] = (int)(npos[t] * rb);
                fastdiv_setup((unsigned)rb, &a.pm_pps_mul[t], &a.pm_pps_sh[t]);
                blk += (int)ns;
            }
            a.pm_blk0[T] = blk;
            if (blk > WGRAD_PM_MAX_BLOCKS) return (int)hipErrorInvalidValue;      // (budget = 256 / (tiles * groups) at most)
            {   // interleave the taps: sort (tap, split) by the split's relative position (s + 0.5) / ns_t, ties by tap
                int ns[CONV_MAX_TAPS], nxt[CONV_MAX_TAPS];
                for (int t = 0; t < T; ++t) { ns[t] = a.pm_blk0[t + 1] - a.pm_blk0[t]; nxt[t] = 0; }
                for (int j = 0; j < blk; ++j) {
                    int best = -1;
                    for (int t = 0; t < T; ++t) {
                        if (nxt[t] >= ns[t]) continue;
                        // (2 nxt + 1) / (2 ns) compared by cross-multiplication
                        if (best < 0 || (long)(2 * nxt[t] + 1) * ns[best] < (long)(2 * nxt[best] + 1) * ns[t]) best = t;
                    }
                    a.pm_tap[j] = (unsigned char)best;
                    a.pm_split[j] = (unsigned char)nxt[best]++;
                }
            }
            fastdiv_setup((unsigned)B, &a.b_mul, &a.b_sh);
            a.pm_inc_ok = ((long)groups * B * GH * GW * Cout < 0x7fffffffL && (long)groups * B * XH * XW * Cin < 0x7fffffffL) ? 1 : 0;
            int maxns = 0;
            bool every = true;
            for (int t = 0; t < T; ++t) {
                const int ns = a.pm_blk0[t + 1] - a.pm_blk0[t];
                maxns = std::max(maxns, ns);
                every = every && ns >= 1;
            }
            a.nslots = maxns;
            if (mode == 1) { if (nslots_out) *nslots_out = (slots_ok && every) ? maxns : 0; return 0; }
            if (mode == 2 && (!every || dw_gs != (int64_t)maxns * a.split_stride)) return (int)hipErrorInvalidValue;
            dim3 grid(tiles, groups, blk);
            if (piped) return loft_launch_conv_wgrad_stream(a, grid, true, (hipStream_t)stream);
            if (big) hipLaunchKernelGGL((conv_wgrad_kernel<256, 8, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((conv_wgrad_kernel<128, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
            LOFT_LAUNCH_CHECK();
            return 0;
        }
        return mode == 2 ? (int)hipErrorInvalidValue : 0;     // (no tap has a valid row: dw stays as the caller left it)
    }
    splits = (int)((M + pps - 1) / pps);
    a.nslots = splits;
    if (mode == 1) { if (nslots_out) *nslots_out = slots_ok ? splits : 0; return 0; }
    if (mode == 2 && dw_gs != (int64_t)splits * a.split_stride) return (int)hipErrorInvalidValue;
    dim3 grid(tiles, T * groups, splits);
    if (piped) return loft_launch_conv_wgrad_stream(a, grid, false, (hipStream_t)stream);
    if (narrow) {
        const bool samesize = gos == 1 && ss == 1 && GH == OH && GW == OW && XH == OH && XW == OW;
        bool same = samesize && OW >= 64;
        for (int t = 0; t < T; ++t) same = same && a.goy[t] == 0 && a.gox[t] == 0;
        const bool dense = samesize && T == 1 && a.goy[0] == 0 && a.gox[0] == 0 && a.dy[0] == 0 && a.dx[0] == 0;
        if (dense) hipLaunchKernelGGL(conv_wgrad64_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (same) hipLaunchKernelGGL(conv_wgrad64_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(conv_wgrad64_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    else if (big)
        hipLaunchKernelGGL((conv_wgrad_kernel<256, 8>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (variant != LOFT_WGRAD_T128)
        return loft_launch_conv_wgrad_ring(a, grid, (hipStream_t)stream);          // four-stage ring (conv_wgrad_pipe.hip)
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_conv_wgrad_bf16_v(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH,
                                       int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                                       const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                                       const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                                       int splits, float* db, int db_tap, int variant, void* stream) {
    return wgrad_impl(g, x, dw, zero_page, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host, gox_host, dy_host, dx_host,
                      wt_host, groups, g_gs, x_gs, dw_gs, splits, db, db_tap, variant, stream, 0, nullptr);
}

// The same contraction on OPERAND PLANES (the fp32 parity mode on the 16-bit matrix cores; see loft_hip.h): g / x hold the planes
// of the fp32 gradient / activation (loft_split_planes_f32), term p multiplies G plane gpl[p] with X plane xpl[p]; every term adds
// into dw (zeroed by the caller) through the split-K atomics, scaled by 1 / (scale_g * scale_x) when the planes are scaled.
LOFT_EXPORT int loft_conv_wgrad_planes(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH,
                                       int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                                       const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                                       const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                                       int nterms, const int* gpl_host, const int* xpl_host, int64_t g_ps, int64_t x_ps,
                                       const float* amax_g, const float* amax_x, float* db, int db_tap, void* stream) {
    if ((Cin % 128) || (Cout % 128) || (amax_g == nullptr) != (amax_x == nullptr)) return (int)hipErrorInvalidValue;
    WgradPlanes pl{nterms, gpl_host, xpl_host, g_ps, x_ps, amax_g, amax_x};
    return wgrad_impl(g, x, dw, zero_page, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host, gox_host, dy_host, dx_host,
                      wt_host, groups, g_gs, x_gs, dw_gs, 0, db, db_tap, LOFT_WGRAD_AUTO, stream, 0, nullptr, &pl);
}

LOFT_EXPORT int loft_conv_wgrad_slots(int B, int GH, int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss,
                                      int T, const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                                      const int* wt_host, int groups, int splits, int variant) {
    int n = 0;
    const int e = wgrad_impl(nullptr, nullptr, nullptr, nullptr, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host,
                             gox_host, dy_host, dx_host, wt_host, groups, 0, 0, 0, splits, nullptr, -1, variant, nullptr, 1, &n);
    return e ? -e : n;
}

LOFT_EXPORT int loft_conv_wgrad_bf16_slots(const void* g, const void* x, float* dw_slots, const void* zero_page, int B, int GH,
                                           int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                                           const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                                           const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int nslots,
                                           int splits, float* db, int db_tap, int variant, void* stream) {
    return wgrad_impl(g, x, dw_slots, zero_page, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host, gox_host, dy_host,
                      dx_host, wt_host, groups, g_gs, x_gs, (int64_t)nslots * T * Cout * Cin, splits, db, db_tap, variant, stream, 2,
                      nullptr);
}

LOFT_EXPORT int loft_conv_wgrad_bf16(const void* g, const void* x, float* dw, const void* zero_page, int B, int GH,
                                     int GW, int Cout, int XH, int XW, int Cin, int OH, int OW, int gos, int ss, int T,
                                     const int* goy_host, const int* gox_host, const int* dy_host, const int* dx_host,
                                     const int* wt_host, int groups, int64_t g_gs, int64_t x_gs, int64_t dw_gs,
                                     int splits, float* db, int db_tap, void* stream) {
    return loft_conv_wgrad_bf16_v(g, x, dw, zero_page, B, GH, GW, Cout, XH, XW, Cin, OH, OW, gos, ss, T, goy_host, gox_host, dy_host,
                                  dx_host, wt_host, groups, g_gs, x_gs, dw_gs, splits, db, db_tap, LOFT_WGRAD_AUTO, stream);
}

// =====================================================================================
// Stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + folded frozen BN + ReLU on MFMA
// (mmdet/models/backbones/resnet.py:628-630).  K = 3*7*7 = 147 is far too shallow per tap for the tap
// kernel, so the A tile is an *LDS-only* im2col: each workgroup gathers its 8x16 output pixels' patches
// straight from the fp32 NCHW image into the swizzled bf16 LDS layout (k = c*49 + r*7 + s, zero padded
// to 192 = 3 chunks of 64) -- nothing im2col-shaped ever touches HBM.  wgt: bf16 [64][192] with the BN
// scale folded in, bias = BN shift.  out: bf16 NHWC [B, H/2, W/2, 64].
// =====================================================================================
constexpr int STEM_PR = 21, STEM_PC = 37, STEM_PP = 40;      // input patch of an 8 x 16 output tile: rows, cols, row pitch

// k = c*49 + r*7 + s columns [KH*32, KH*32+32) of im2col chunk CHUNK for output pixel m, from the LDS patch (all offsets are
// compile-time constants: no divisions, LDS immediates)
template <int CHUNK, int KH>
__device__ __forceinline__ void stem_gather(const float* patch, char* abuf, int m) {
    const float* pm = patch + 2 * (m >> 4) * STEM_PP + 2 * (m & 15);
#pragma unroll
    for (int q8 = 0; q8 < 4; ++q8) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            constexpr int kb = CHUNK * 64 + KH * 32;
            const int k = kb + q8 * 8 + e;
            xv[e] = 0.f;
            if (k < 147) {
                const int c = k / 49, rs = k - c * 49, r = rs / 7, s2 = rs - r * 7;
                xv[e] = pm[(c * STEM_PR + r) * STEM_PP + s2];
            }
        }
        // (hardware conversion, two values per instruction: the software form's NaN branch was ten instructions per element)
        const int q = KH * 4 + q8;
        *reinterpret_cast<uint4*>(abuf + m * 128 + swz(m, q) * 16) = pack8_16(xv);
    }
}

__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wgt,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out, int H, int W,
                                                        int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 128 + 64 * 128];  // A chunk [128][64] + B chunk [64][64]
    __shared__ float patch[3 * STEM_PR * STEM_PP];                           // the tile's fp32 input window, zero padded
    char* abuf = lds;
    char* bbuf = lds + 128 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, oy0 = blockIdx.y * 8, ox0 = blockIdx.x * 16;
    const int m = tid & 127, kh = tid >> 7;
    const float* ib = img + (long)b * 3 * H * W;
    // ---- the 3 x 21 x 37 input window once, row segments coalesced (the previous form gathered every im2col entry from
    // global memory: 73 scattered 4-byte loads per thread; measured 470 us -> see DESIGN.md)
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    for (int i = tid; i < 3 * STEM_PR * STEM_PP; i += 256) {
        const int col = i % STEM_PP, rr = i / STEM_PP, r = rr % STEM_PR, c = rr / STEM_PR;
        const int iy = iy0 + r, ix = ix0 + col;
        float x = 0.f;
        if (col < STEM_PC && iy >= 0 && iy < H && ix >= 0 && ix < W) x = ib[((long)c * H + iy) * W + ix];
        patch[i] = x;
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int frow = lane & 31, fq = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int chunk = 0; chunk < 3; ++chunk) {
        if (chunk) __syncthreads();
        // ---- A: this thread fills k = chunk*64 + kh*32 + [0,32) of row m (rows of pixels beyond the map are never stored)
        if (kh == 0) {
            if (chunk == 0) stem_gather<0, 0>(patch, abuf, m); else if (chunk == 1) stem_gather<1, 0>(patch, abuf, m); else stem_gather<2, 0>(patch, abuf, m);
        } else {
            if (chunk == 0) stem_gather<0, 1>(patch, abuf, m); else if (chunk == 1) stem_gather<1, 1>(patch, abuf, m); else stem_gather<2, 1>(patch, abuf, m);
        }
        // ---- B chunk: 64 rows x 8 sixteen-byte pieces = 512 pieces, 2 per thread
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = tid + i * 256, row = piece >> 3, q = piece & 7;
            *reinterpret_cast<uint4*>(bbuf + row * 128 + swz(row, q) * 16) =
                *reinterpret_cast<const uint4*>(wgt + row * 192 + chunk * 64 + q * 8);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks * 2 + fq;
            const int arow = wave * 32 + frow;
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(abuf + arow * 128 + swz(arow, q) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int brow = i * 32 + frow;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(bbuf + brow * 128 + swz(brow, q) * 16);
                acc[i] = LOFT_MFMA_32x32x16(wf, xf, acc[i]);
            }
        }
    }
    const int mo = wave * 32 + frow;
    const int py = oy0 + (mo >> 4), px = ox0 + (mo & 15);
    if (py >= Ho || px >= Wo) return;
    bf16_t* op = out + (((long)b * Ho + py) * Wo + px) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = i * 32 + 8 * gq + 4 * fq;
            const float4 bv = *reinterpret_cast<const float4*>(bias + n);
            float v[4] = {fmaxf(acc[i][gq * 4 + 0] + bv.x, 0.f), fmaxf(acc[i][gq * 4 + 1] + bv.y, 0.f),
                          fmaxf(acc[i][gq * 4 + 2] + bv.z, 0.f), fmaxf(acc[i][gq * 4 + 3] + bv.w, 0.f)};
            st4(op + n, v);
        }
}

LOFT_EXPORT int loft_stem7x7_mfma(const float* img, const void* wgt_packed, const float* bias, void* out, int B, int H, int W,
                                  void* stream) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid(loft_cdiv(Wo, 16), loft_cdiv(Ho, 8), B);
    hipLaunchKernelGGL(stem_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, (const bf16_t*)wgt_packed, bias,
                       (bf16_t*)out, H, W, Ho, Wo);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================
// fp32 parity mode (forward / data-gradient only).  Same tap-convolution contract as loft_conv_tap_bf16, but every
// operand is fp32 and the contraction runs on v_mfma_f32_32x32x2_f32 -- bit-for-bit an fp32 fmaf chain (exact fp32,
// 1/16 of the bf16 MFMA rate).  It exists so that inference outputs (boxes, masks, offsets) can be compared with the
// fp32 CPU oracle at the north-star tolerance of 1e-3; it is not a performance path (its b32 LDS reads are fully
// bank-conflicted by construction).  128x128 tile, K-step = 32 channels (the same 128-byte LDS rows).
// =====================================================================================
struct ConvArgsF32 {
    const float* src; const float* wgt; const float* bias; const float* residual; const float* mask; float* out;
    const float* zero_page;
    int B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo_y, oo_x, ss, T;
    int dy[CONV_MAX_TAPS], dx[CONV_MAX_TAPS], wt[CONV_MAX_TAPS];
    int relu, accumulate;
    long src_gs, wgt_gs, out_gs, bias_gs;
    int M;
};

// SPLIT (the mode's default since round 4, LOFT_F32_SPLIT3): the same staging, but every fp32 operand element is split on its way
// into the matrix core into two bf16 values, x = hi + lo + O(2^-18 |x|) (hi = RNE(x), lo = RNE(x - hi): 16 mantissa bits), and a
// product w * x becomes three v_mfma_f32_32x32x16_bf16 terms, wh*xh + wh*xl + wl*xh (the dropped wl*xl is 2^-18 of the
// product), accumulated in fp32: per-product error <= ~1e-5 |w x| with random sign -- 1e-6 of an output's scale after the
// K-sum -- at 16 / 3 of the fp32 MFMA rate, with 16-byte fragment reads instead of conflicted 4-byte ones.  The explicit bf16
// types keep this path identical in the library's f16 build (an f16 split would overflow on fp32-range values).
typedef __attribute__((ext_vector_type(8))) __bf16 xbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 xbf16x2;
__device__ __forceinline__ void f32_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const loft_f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xbf16x2));
    const loft_f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, xbf16x2));
}
__device__ __forceinline__ void f32_split8(const float4 p, const float4 q, xbf16x8& hi, xbf16x8& lo) {
    uint4 h, l;
    f32_split2(p.x, p.y, h.x, l.x); f32_split2(p.z, p.w, h.y, l.y); f32_split2(q.x, q.y, h.z, l.z); f32_split2(q.z, q.w, h.w, l.w);
    hi = __builtin_bit_cast(xbf16x8, h); lo = __builtin_bit_cast(xbf16x8, l);
}
// three bf16 per fp32 (24 mantissa bits: x = hi + mid + lo to fp32 accuracy) -- LOFT_F32_SPLIT6
__device__ __forceinline__ void f32_split3x2(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const loft_f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xbf16x2));
    const loft_f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, xbf16x2));
    const loft_f32x2 r2 = {r[0] - __uint_as_float(mid << 16), r[1] - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, xbf16x2));
}
__device__ __forceinline__ void f32_split8x3(const float4 p, const float4 q, xbf16x8& hi, xbf16x8& mid, xbf16x8& lo) {
    uint4 h, m, l;
    f32_split3x2(p.x, p.y, h.x, m.x, l.x); f32_split3x2(p.z, p.w, h.y, m.y, l.y);
    f32_split3x2(q.x, q.y, h.z, m.z, l.z); f32_split3x2(q.z, q.w, h.w, m.w, l.w);
    hi = __builtin_bit_cast(xbf16x8, h); mid = __builtin_bit_cast(xbf16x8, m); lo = __builtin_bit_cast(xbf16x8, l);
}

template <int SPLIT>      // 0: exact fp32 MFMA, 2: two bf16 per operand (3 terms), 3: three bf16 per operand (6 terms)
__global__ __launch_bounds__(256) void conv_tap_f32_kernel(const ConvArgsF32 a) {
    constexpr int BM = 128, BN = 128, BKE = 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    __shared__ __attribute__((aligned(16))) char lds[2 * (A_BYTES + B_BYTES)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, g = blockIdx.z;
    const float* src = a.src + (long)g * a.src_gs;
    const float* wgt = a.wgt + (long)g * a.wgt_gs;
    const int lrow = lane >> 3, lchunk = lane & 7;
    int a_base[4], a_y[4], a_x[4], a_c[4];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + wave * 8 + lrow;
        const int m = m0 + row;
        a_c[i] = swz(row, lchunk) * 4;
        if (m < a.M) {
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            a_base[i] = b * a.IH * a.IW; a_y[i] = oy * a.ss; a_x[i] = ox * a.ss;
        } else { a_base[i] = 0; a_y[i] = -100000; a_x[i] = -100000; }
    }
    long b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + wave * 8 + lrow;
        const int n = n0 + row;
        b_off[i] = (n < a.Cout) ? ((long)n * a.Cin + swz(row, lchunk) * 4) : -1;
    }
    const int kchunks = a.Cin / BKE;
    const int nk = a.T * kchunks;
    auto stage = [&](int kk, int buf) {
        const int t = kk / kchunks, c0 = (kk - t * kchunks) * BKE;
        const int dy = a.dy[t], dx = a.dx[t];
        char* abuf = lds + buf * (A_BYTES + B_BYTES);
        char* bbuf = abuf + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_y[i] + dy, ix = a_x[i] + dx;
            const bool ok = (iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW);
            const float* p = ok ? src + ((long)(a_base[i] + iy * a.IW + ix) * a.Cin + c0 + a_c[i]) : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(abuf + (i * 32 + wave * 8) * 128), 16, 0, 0);
        }
        const float* wt = wgt + (long)a.wt[t] * a.Cout * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* p = (b_off[i] >= 0) ? wt + b_off[i] : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(bbuf + (i * 32 + wave * 8) * 128), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fq = lane >> 5;
    stage(0, 0);
    for (int kk = 0; kk < nk; ++kk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kk + 1 < nk) stage(kk + 1, (kk + 1) & 1);
        const char* abuf = lds + (kk & 1) * (A_BYTES + B_BYTES);
        const char* bbuf = abuf + A_BYTES;
        if constexpr (SPLIT == 3) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int q0 = ks * 4 + fq * 2;
                xbf16x8 wh[2], wm_[2], wl[2], xh[2], xm[2], xl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wn * 64 + i * 32 + frow;
                    f32_split8x3(*reinterpret_cast<const float4*>(bbuf + row * 128 + swz(row, q0) * 16),
                                 *reinterpret_cast<const float4*>(bbuf + row * 128 + swz(row, q0 + 1) * 16), wh[i], wm_[i], wl[i]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wm * 64 + j * 32 + frow;
                    f32_split8x3(*reinterpret_cast<const float4*>(abuf + row * 128 + swz(row, q0) * 16),
                                 *reinterpret_cast<const float4*>(abuf + row * 128 + swz(row, q0 + 1) * 16), xh[j], xm[j], xl[j]);
                }
                // smallest terms first: hi*lo, lo*hi, mid*mid (2^-16), then hi*mid, mid*hi (2^-8), then hi*hi
#define F32_TERM(A_, B_)                                                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0)
                F32_TERM(wh, xl); F32_TERM(wl, xh); F32_TERM(wm_, xm); F32_TERM(wh, xm); F32_TERM(wm_, xh); F32_TERM(wh, xh);
#undef F32_TERM
            }
        } else if constexpr (SPLIT == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                     // two 16-channel MFMA steps per 32-channel stage
                const int q0 = ks * 4 + fq * 2;                  // this lane's 8 channels = logical chunks q0, q0 + 1
                xbf16x8 wh[2], wl[2], xh[2], xl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wn * 64 + i * 32 + frow;
                    f32_split8(*reinterpret_cast<const float4*>(bbuf + row * 128 + swz(row, q0) * 16),
                               *reinterpret_cast<const float4*>(bbuf + row * 128 + swz(row, q0 + 1) * 16), wh[i], wl[i]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wm * 64 + j * 32 + frow;
                    f32_split8(*reinterpret_cast<const float4*>(abuf + row * 128 + swz(row, q0) * 16),
                               *reinterpret_cast<const float4*>(abuf + row * 128 + swz(row, q0 + 1) * 16), xh[j], xl[j]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[i], xh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xh[j], acc[i][j], 0, 0, 0);
            }
        } else
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int k = ks * 2 + fq, q = k >> 2, e = k & 3;
            float wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wn * 64 + i * 32 + frow;
                wf[i] = *reinterpret_cast<const float*>(bbuf + row * 128 + swz(row, q) * 16 + e * 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wm * 64 + j * 32 + frow;
                xf[j] = *reinterpret_cast<const float*>(abuf + row * 128 + swz(row, q) * 16 + e * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    const float* bias = a.bias ? a.bias + (long)g * a.bias_gs : nullptr;
    const long out_g = (long)g * a.out_gs;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + frow;
        if (m >= a.M) continue;
        const int b = m / ohw, rem = m - b * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        const long opix = ((long)b * a.OHf + oy * a.os + a.oo_y) * a.OWf + ox * a.os + a.oo_x;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * gq + 4 * fq;
                if (n >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][gq * 4 + e];
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                const long o = out_g + opix * a.Cout + n;
                if (a.residual) {
                    float rv[4];
                    ld4(a.residual + o, rv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (a.mask) {
                    float mv[4];
                    ld4(a.mask + o, mv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                }
                if (a.accumulate) {
                    float ov[4];
                    ld4(a.out + o, ov);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += ov[e];
                }
                st4(a.out + o, v);
            }
    }
}

LOFT_EXPORT int loft_conv_tap_f32_v(const float* src, const float* wgt, const float* bias, const float* residual,
                                    const float* relu_mask, float* out, const void* zero_page, int B, int IH, int IW, int Cin,
                                    int Cout, int OH, int OW, int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T,
                                    const int* dy_host, const int* dx_host, const int* wt_host, int relu, int accumulate,
                                    int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs, int variant,
                                    void* stream) {
    if (variant != LOFT_F32_SPLIT6 && variant != LOFT_F32_SPLIT3 && variant != LOFT_F32_EXACT) return (int)hipErrorInvalidValue;
    if (T < 1 || T > CONV_MAX_TAPS || (Cin % 32) || (Cout % 4) || groups < 1) return (int)hipErrorInvalidValue;
    ConvArgsF32 a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.residual = residual; a.mask = relu_mask; a.out = out;
    a.zero_page = (const float*)zero_page;
    a.B = B; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.OHf = OHf; a.OWf = OWf;
    a.os = os; a.oo_y = oo_y; a.oo_x = oo_x; a.ss = ss; a.T = T;
    for (int t = 0; t < T; ++t) { a.dy[t] = dy_host[t]; a.dx[t] = dx_host[t]; a.wt[t] = wt_host[t]; }
    a.relu = relu; a.accumulate = accumulate;
    a.src_gs = src_gs; a.wgt_gs = wgt_gs; a.out_gs = out_gs; a.bias_gs = bias_gs;
    const long M = (long)B * OH * OW;
    if (M <= 0) return 0;
    if (M > 0x7fffffffL) return (int)hipErrorInvalidValue;
    a.M = (int)M;
    dim3 grid(loft_cdiv(M, 128), loft_cdiv(Cout, 128), groups);
    if (variant == LOFT_F32_EXACT) hipLaunchKernelGGL(conv_tap_f32_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (variant == LOFT_F32_SPLIT3) hipLaunchKernelGGL(conv_tap_f32_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_tap_f32_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

LOFT_EXPORT int loft_conv_tap_f32(const float* src, const float* wgt, const float* bias, const float* residual,
                                  const float* relu_mask, float* out, const void* zero_page, int B, int IH, int IW, int Cin,
                                  int Cout, int OH, int OW, int OHf, int OWf, int os, int oo_y, int oo_x, int ss, int T,
                                  const int* dy_host, const int* dx_host, const int* wt_host, int relu, int accumulate,
                                  int groups, int64_t src_gs, int64_t wgt_gs, int64_t out_gs, int64_t bias_gs, void* stream) {
    return loft_conv_tap_f32_v(src, wgt, bias, residual, relu_mask, out, zero_page, B, IH, IW, Cin, Cout, OH, OW, OHf, OWf, os, oo_y,
                               oo_x, ss, T, dy_host, dx_host, wt_host, relu, accumulate, groups, src_gs, wgt_gs, out_gs, bias_gs,
                               LOFT_F32_SPLIT6, stream);
}
