// conv_pipe.hip -- the deep-K form of the NHWC tap convolution (contract and epilogue: conv_mfma.hip / conv_tap.h) as a
// software-pipelined 256 x 256 x 64 kernel for gfx950.
//
// Reference call sites served: every 3x3 / FC contraction with K = T*Cin >= 512 and Cout % 256 == 0 -- FPN output convs
// (necks/fpn.py:170-199), the RPN 3x3 (dense_heads/rpn_head.py:38-44), bottleneck 3x3s (backbones/resnet.py:266-298), the
// mask head (mask_heads/fcn_mask_head.py:118-126), the FOA branches (attribute_heads/offset_head_expand_feature.py:134-161)
// and the shared FCs, forward and data gradient.
//
// Why a second kernel.  conv_tap_kernel<256,256,...> runs all eight waves in lockstep: per K-tile every wave issues its eight
// global->LDS copies, waits at the barrier, reads 24 fragments, issues 32 MFMAs.  While the copies are being issued (60-180
// cycles each, plus the per-row address selects) and while the barrier drains, the matrix pipe of every SIMD idles: 33 % of the
// bf16 MFMA peak on the best shape.  Here the eight waves form TWO GROUPS of four (one wave of each group per SIMD) that run
// the same instruction stream ONE BARRIER APART: a K-tile is cut into four phases, each phase = a "load" half (fragment reads of
// one accumulator quadrant + ONE half-tile of global->LDS copies for a later K-tile: 2 copies per thread) and an "MFMA" half
// (the quadrant's 8 v_mfma_f32_32x32x16_bf16), separated by raw s_barriers.  Because of the one-barrier stagger, in every slot
// one wave of a SIMD issues MFMAs while its partner reads fragments / issues copies.  Copies stay in flight across barriers and
// are retired by COUNTED s_waitcnt vmcnt(N) (never 0 in the steady state), placed one slot before the first read of the data.
//
// LDS (128 KiB, ONE array -- a second __shared__ object makes hipcc fence every ds_read behind all LDS-DMA):
//   [W buf0 32K][W buf1 32K][X buf0 32K][X buf1 32K]; a tile = 256 rows x 128 B (64 bf16 of one K-tile), 16-byte chunk q of
//   row r at slot q ^ ((r>>1)&7) (applied on the copy's per-lane SOURCE address and on the ds_read_b128 address).
//   Half-tile pieces (16 KiB = 512 threads x 2 copies): W0/W1 = weight rows 0-127 / 128-255, X0/X1 = pixel rows 0-127 / 128-255.
//   Wave (wm, wn): pixels [128 wm, +128) = X piece wm only; couts [64 wn, +64) = W piece wn>>1 only.
//
// Schedule.  Slot s = the code between barrier s-1 and barrier s.  Group 0 (wm = 0) runs L(t,p) in slot 8t+2p and M(t,p) in
// 8t+2p+1; group 1 (wm = 1) one slot later.  Per K-tile t (buffer t&1), quadrants (p0/p1 = first/second 64 pixels, c0/c1 =
// first/second 32 couts of the wave tile):
//   L0: read X p0 (8 x b128), W c0 (4)      + issue W1(t+1)      M0: acc[c0][p0] (8 MFMA)
//   L1: read W c1 (4)                       + issue X0(t+1)      M1: acc[c1][p0]
//   L2: read X p1 (8, same registers)       + issue X1(t+1)      M2: acc[c1][p1]
//   L3: (no reads) advance the tap/channel state, issue W0(t+2), s_waitcnt vmcnt(4)   -> retires X0(t+1) and older
//   M3: acc[c0][p1] (W c0 still in registers),                  s_waitcnt vmcnt(2)   -> retires X1(t+1)
// RAW (a piece is read only after EVERY issuing thread's counted wait and one more barrier): W1(t+1) issued in slots 8t/8t+1,
//   X0(t+1) in 8t+2/3 -- retired in slots 8t+6/7, first read in slot 8t+8; X1(t+1) issued 8t+4/5, retired 8t+7/8, first read (by
//   group 1 only) in 8t+9; W0(t+2) issued 8t+6/7, retired 8t+14/15, first read 8t+16.
// WAR (a piece is overwritten no earlier than two slots after the last ds_read of its previous contents was ISSUED -- the reads
//   are complete after the lgkmcnt wait that opens the reader's next slot): W(t-1) last read in slot 8t-5, X0(t-1) 8t-4,
//   X1(t-1) 8t-3, all before slot 8t; W0(t) last read in slot 8t+3 (group 1's L1), overwritten from slot 8t+6.
// Roofline: MFMA (dense bf16 2.5 PFLOP/s); algorithmic work 2*M*Cout*Cin*T FLOP per launch.
#include "conv_tap.h"
#include <type_traits>
#include "../../include/loft_hip.h"

namespace {
constexpr int PW_OFF = 0, PX_OFF = 65536, PBUF = 32768, PLDS = 131072;
}

#define PIPE_SB() __builtin_amdgcn_sched_barrier(0)
#define PIPE_BARRIER()                      \
    do {                                    \
        PIPE_SB();                          \
        __builtin_amdgcn_s_barrier();       \
        PIPE_SB();                          \
    } while (0)


// Epilogue of the 256 x 256 kernels for bf16 outputs: bias + residual + ReLU + ReLU-backward mask as conv_epilogue, but every
// HBM access of the tile is a whole 512-byte output-pixel row (16 bytes per lane, 32 lanes per row): the result tile (128 KiB
// of bf16 = the LDS the K loop just released) is collected in LDS and copied out row by row; residual / mask tiles come in
// the same way.  The direct form stores 8 bytes per lane to 64 different rows per instruction, 32 instructions per wave --
// measured 25-27k cycles per workgroup on the 8 x 256^2 P2 conv against 108k for its whole K loop; this form: see DESIGN.md.
// LDS image: row r (tile pixel row) x 32 chunks of 16 B, logical chunk c at position c ^ (r & 31).
// row m of the launch -> (image / RoI b, pixel index rem inside its map); b >= a.B marks a padding row (blocked pixel-major order)
__device__ __forceinline__ void pipe_row_decode(const ConvArgs& a, int m, int ohw, int& b, int& rem) {
    if (a.pixmajor) {
        const int seg = fastdiv(m, a.pms_mul, a.pms_sh), blk = fastdiv(seg, a.pmp_mul, a.pmp_sh);
        rem = seg - blk * a.pm_P;
        b = blk * a.pm_S + (m - seg * a.pm_S);
    } else { b = fastdiv(m, a.ohw_mul, a.ohw_sh); rem = m - b * ohw; }
}
__device__ __forceinline__ const bf16_t* pipe_row_ptr(const ConvArgs& a, const bf16_t* base, long out_g, int m, int n0, int ohw, int ooy,
                                                      int oox) {
    int b, rem;
    pipe_row_decode(a, m, ohw, b, rem);
    if (b >= a.B) return nullptr;
    const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
    const long opix = ((long)b * a.OHf + oy * a.os + ooy) * a.OWf + ox * a.os + oox;
    return base + out_g + opix * a.Cout + n0;
}
__device__ __forceinline__ const bf16_t* pipe_row_ptr(const ConvArgs& a, const bf16_t* base, long out_g, int m, int n0, int ohw) {
    return pipe_row_ptr(a, base, out_g, m, n0, ohw, a.oo_y, a.oo_x);
}

// Wave-uniform description of where the rows of ONE tile live in an NHWC map (round 4).  The per-row decode above costs three
// multiply-high divisions and a chain of 64-bit multiplies per row AND -- what the barrier-level trace of round 4 showed to be the
// real price -- a dozen kernarg (SMEM) reloads per row, each behind its own s_waitcnt lgkmcnt(0): hipcc does not keep ~15 ConvArgs
// scalars live across the K loop, so the 16 row stores of the epilogue ran ~120 instructions and 6-9 scalar-memory round trips
// apiece (10.9k of a tile's 147k cycles on the FOA maps, 8.7k more in the set-up).  Both layouts the RoI heads and the backbone
// use are piecewise LINEAR in the tile row r:
//   pixel-major RoI blocks: rows [0, rs1) are RoIs b00, b00+1, .. at pixel position pos0, rows [rs1, BM) RoIs b10, .. at pos1
//                           (a tile spans at most two segments because pm_S >= 256 >= BM) -- element offset = off_k + idx * stride
//                           with stride = (map pixels) * channels;
//   dense maps (output pixel index == m): one segment, stride = channels.
// Six scalars per tile, computed once on the scalar unit; a row costs a compare, two selects and one 64-bit multiply-add.
struct TileRows {
    int lin;                 // 0: neither layout (strided / offset outputs): the per-row decode above
    int rs1, nv0, nv1;       // first row of the second segment (>= BM: none); valid rows per segment
    long off0, off1, stride; // element offsets
};
__device__ __forceinline__ int pipe_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// pixel-major tile -> its two segments: first RoI and pixel position of each
__device__ __forceinline__ void pipe_pm_tile(const ConvArgs& a, int m0, int& rs1, int& b00, int& b10, int& pos0, int& pos1) {
    const int seg0 = fastdiv(m0, a.pms_mul, a.pms_sh), blk0 = fastdiv(seg0, a.pmp_mul, a.pmp_sh);
    pos0 = seg0 - blk0 * a.pm_P;
    const int r0 = m0 - seg0 * a.pm_S;
    b00 = blk0 * a.pm_S + r0;
    rs1 = a.pm_S - r0;
    const bool wrap = pos0 + 1 == a.pm_P;
    pos1 = wrap ? 0 : pos0 + 1;
    b10 = (wrap ? blk0 + 1 : blk0) * a.pm_S;
}
// rows of the tile in a map of [B][H][W][C] whose pixel (oy, ox) of output position `pos` is (oy * st + o_y, ox * st + o_x)
template <int BM>
__device__ __forceinline__ TileRows pipe_tile_rows(const ConvArgs& a, int m0, bool dense, int H, int W, int C, int st, int o_y, int o_x) {
    TileRows t;
    t.lin = 1; t.rs1 = BM; t.nv1 = 0; t.off1 = 0;
    if (a.pixmajor) {
        int rs1, b00, b10, pos0, pos1;
        pipe_pm_tile(a, m0, rs1, b00, b10, pos0, pos1);
        const int oy0 = fastdiv(pos0, a.ow_mul, a.ow_sh), ox0 = pos0 - oy0 * a.OW;
        const int oy1 = fastdiv(pos1, a.ow_mul, a.ow_sh), ox1 = pos1 - oy1 * a.OW;
        const int rows = pipe_clamp(a.M - m0, 0, BM);
        t.rs1 = rs1 < BM ? rs1 : BM;
        t.nv0 = pipe_clamp(a.B - b00, 0, t.rs1 < rows ? t.rs1 : rows);
        t.nv1 = pipe_clamp(a.B - b10, 0, rows - t.rs1 > 0 ? rows - t.rs1 : 0);
        t.off0 = (((long)b00 * H + oy0 * st + o_y) * W + ox0 * st + o_x) * C;
        t.off1 = (((long)b10 * H + oy1 * st + o_y) * W + ox1 * st + o_x) * C;
        t.stride = (long)H * W * C;
    } else if (dense) {
        t.nv0 = pipe_clamp(a.M - m0, 0, BM);
        t.off0 = (long)m0 * C;
        t.stride = C;
    } else {
        t.lin = 0; t.nv0 = 0; t.stride = 0; t.off0 = 0;
    }
    return t;
}
__device__ __forceinline__ long pipe_row_off(const TileRows& t, int r, bool& ok) {
    const bool k = r >= t.rs1;
    const int idx = k ? r - t.rs1 : r;
    ok = idx < (k ? t.nv1 : t.nv0);
    return (k ? t.off1 : t.off0) + (long)idx * t.stride;
}

// bias (+ residual) + ReLU -> bf16 -> LDS -> whole rows out; the ReLU-backward mask (data-gradient launches) is applied in the
// copy-out pass on the packed bf16 values against 16-byte mask loads with the address pattern of the stores -- the accumulators
// are dead by then.  RES: the residual tile (bottleneck shortcut, resnet.py:294-296; shortcut gradient of a fused block) comes in
// through LDS with whole-row copies first; every lane reads its 8-byte pieces from there, adds in fp32 and overwrites them with
// the result in place.
// NW = 32-cout blocks per wave: 2 -> 256 couts per tile (512-byte rows, 32 chunks), 1 -> 128 couts (256-byte rows, 16 chunks).
template <int MJ, int NW>
__device__ __forceinline__ void pipe_stage_in(const ConvArgs& a, const TileRows& tr, const bf16_t* t, long out_g, int m0, int n0, int ohw,
                                              int wave, int lane, char* lds) {
    constexpr int RW = 8 * MJ;                    // tile rows copied by one wave (8 waves cover 64 MJ rows)
    constexpr int CPR = 16 * NW, RPI = 64 / CPR;  // 16-byte chunks per row; rows per wave-level copy
    const bf16_t* tb = t + out_g + n0;
#pragma unroll 1
    for (int it = 0; it < RW / RPI; ++it) {
        const int r = wave * RW + it * RPI + lane / CPR;
        const int c = (lane & (CPR - 1)) ^ (r & (CPR - 1));   // the LDS image of a glds is lane-linear: swizzle the SOURCE chunk
        const bf16_t* p = a.zero_page;
        if (tr.lin) {
            bool ok;
            const long off = pipe_row_off(tr, r, ok);
            if (ok) p = tb + off + c * 8;
        } else {
            const int m = m0 + r;
            if (m < a.M) {
                const bf16_t* rp = pipe_row_ptr(a, t, out_g, m, n0, ohw);
                if (rp) p = rp + c * 8;
            }
        }
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(lds + (wave * RW + it * RPI) * (CPR * 16)), 16, 0, 0);
    }
}

template <bool RES, int MJ, int NW, typename StampFn>
__device__ __forceinline__ void pipe_epilogue_staged(const ConvArgs& a, const TileRows& tr, f32x16 (&acc)[2][4], char* lds, int g, int m0,
                                                     int n0, int wave, int lane, int ohw, StampFn&& kstamp, int ooy, int oox) {
    constexpr int RW = 8 * MJ;
    constexpr int CPR = 16 * NW, ROWB = CPR * 16, RPI = 64 / CPR;      // chunks per row, bytes per row, rows per wave-level access
    // (ConvArgs block 2 in one go: see the set-up)
    asm volatile("" :: "s"(a.bias), "s"(a.residual), "s"(a.mask), "s"(a.out_gs), "s"(a.bias_gs), "s"(a.relu), "s"(a.out), "s"(a.zero_page));
    const int wm = wave >> 2, wn = wave & 3, frow = lane & 31, fq = lane >> 5;
    const long out_g = (long)g * a.out_gs;
    // (no bias: read zeros -- a branch around the adds makes hipcc keep two copies of the 128 accumulator registers)
    const float* bias = a.bias ? a.bias + (long)g * a.bias_gs + n0 + wn * (32 * NW) + 4 * fq
                               : reinterpret_cast<const float*>(a.zero_page) + 4 * fq;
    // This lane's 8-byte piece (4 couts) of block (i, gq) in tile row r = wm*128 + j*32 + frow is piece p = wn*16 + i*8 + 2*gq + fq,
    // i.e. 16-byte chunk wn*8 + k (k = i*4 + gq) at position chunk ^ (r & 31) = ((wn ^ (frow>>3)) << 3) | (k ^ (frow & 7)):
    // eight per-lane offsets, the row block j is a ds immediate (j * 16 KiB).
    // (NW = 1: the wave's 32 couts are the four chunks wn*4 + gq of a 16-chunk row: position ((wn ^ ((frow>>2)&3)) << 2) | (gq ^ (frow&3)))
    char* rowb = lds + (wm * (32 * MJ) + frow) * ROWB + fq * 8 +
                 (NW == 2 ? (((wn ^ (frow >> 3)) << 3) << 4) : (((wn ^ ((frow >> 2) & 3)) << 2) << 4));
    const int f7 = NW == 2 ? (frow & 7) : (frow & 3);
    const float lo = a.relu ? 0.f : -__builtin_inff();          // ReLU as a branch-free max
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave is done with the K loop's fragments
    kstamp(44);
    // (narrow head, below: its weight fragments are requested HERE -- the K loop's fragment registers are dead -- so that the L2
    //  round trip runs under the staging pass instead of in front of the head's MFMAs)
    bf16x8 hwf[NW == 2 ? 16 : 1];
    bool head_on = false;
    if constexpr (NW == 2) {
        head_on = a.head_w != nullptr && wave < 2 * MJ;
        const bool hrow = frow < a.head_c4;               // (head_w holds head_c4 rows: the MFMA's other rows are zeros)
        const bf16_t* hw = a.head_w + frow * 256 + fq * 8;
        if (head_on) {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hwf[st][e] = 0;
                if (hrow) hwf[st] = *reinterpret_cast<const bf16x8*>(hw + st * 16);
            }
        }
    }
    if constexpr (RES) {
        pipe_stage_in<MJ, NW>(a, tr, a.residual, out_g, m0, n0, ohw, wave, lane, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4 * NW; ++k) {
        const int i = k >> 2, gq = k & 3;
        const float4 bv = *reinterpret_cast<const float4*>(bias + i * 32 + 8 * gq);
        char* q = rowb + ((k ^ f7) << 4);
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            float v[4];
            v[0] = acc[i][j][gq * 4 + 0] + bv.x; v[1] = acc[i][j][gq * 4 + 1] + bv.y;
            v[2] = acc[i][j][gq * 4 + 2] + bv.z; v[3] = acc[i][j][gq * 4 + 3] + bv.w;
            if constexpr (RES) {
                float rv[4];
                ld4(reinterpret_cast<const bf16_t*>(q + j * (32 * ROWB)), rv);
                v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
            }
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            st4(reinterpret_cast<bf16_t*>(q + j * (32 * ROWB)), v);
            if constexpr (RES) PIPE_SB();
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    kstamp(46);
    __syncthreads();
    kstamp(47);
    // ---- a narrow 1x1 head on the finished tile (ConvArgs block 5): head_out[pixel][n] = head_b[n] + sum_c tile[pixel][c] * head_w[n][c]
    // as ONE 32x32 MFMA chain per 32-row block -- A = the head's weights (rows n, 16 bytes per lane per 16-channel step straight from
    // L2: 16 KiB in all), B = the tile rows as they lie in LDS (row frow of the block, logical chunk 2 s + fq at position chunk ^ row:
    // the fragment shape of the K loop), 16 steps over the 256 couts.  The wide map is not read back by a second launch.
    if constexpr (NW == 2) {
        if (head_on) {
            {
                const int r = wave * 32 + frow;
                const char* xrow = lds + r * ROWB;
                f32x16 hacc;
#pragma unroll
                for (int v = 0; v < 16; ++v) hacc[v] = 0.f;
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + (((2 * st + fq) ^ frow) << 4));
                    hacc = LOFT_MFMA_32x32x16(hwf[st], xf, hacc);
                }
                // lane -> pixel (tile row r); hacc[4 gq + e] -> head output 8 gq + 4 fq + e
                long roff = 0;
                bool ok;
                if (tr.lin) roff = pipe_row_off(tr, r, ok);
                else {
                    const int m = m0 + r;
                    ok = m < a.M;
                    if (ok) {
                        const bf16_t* base = reinterpret_cast<const bf16_t*>(a.out);
                        const bf16_t* rp = pipe_row_ptr(a, base, out_g, m, n0, ohw, ooy, oox);
                        ok = rp != nullptr;
                        if (ok) roff = rp - (base + out_g + n0);
                    }
                }
                if (ok) {
                    const int c4 = a.head_c4;
                    float* ho = a.head_out + (roff / a.Cout) * c4;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int nh = 8 * gq + 4 * fq;
                        if (nh < c4) {
                            const float4 hb = *reinterpret_cast<const float4*>(a.head_b + nh);
                            *reinterpret_cast<float4*>(ho + nh) = make_float4(hacc[gq * 4 + 0] + hb.x, hacc[gq * 4 + 1] + hb.y,
                                                                             hacc[gq * 4 + 2] + hb.z, hacc[gq * 4 + 3] + hb.w);
                        }
                    }
                }
            }
        }
    }
    // LDS -> HBM, whole rows.  All 16 row reads first (the accumulators are dead: 64 free registers), then the stores: one
    // lgkmcnt wait for the whole tile instead of a read -> wait -> store chain per row pair.
    const bf16_t* mask = a.mask;
    const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.out) + out_g + n0;
    // keep a bf16 lane where its mask value is > 0: sign bit clear and magnitude bits non-zero
    auto sel = [](unsigned vv, unsigned mm) {
        const unsigned lo16 = ((mm & 0x8000u) == 0u && (mm & 0x7fffu) != 0u) ? 0xffffu : 0u;
        const unsigned hi16 = ((mm & 0x80000000u) == 0u && (mm & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
        return vv & (lo16 | hi16);
    };
    if (tr.lin) {
        // Linear rows: no decode and no kernarg reload per row; a row's address is one of two scalar bases + a 32-bit element
        // offset.  With a mask (data-gradient launches) the mask rows of HALF a tile are requested together before the first is
        // used: the per-row form chained 16 dependent L2 round trips per thread.
        constexpr int NR = RW / RPI, HALF = NR >= 8 ? NR / 2 : NR;
        const bf16_t* ob0 = ob + tr.off0;
        const bf16_t* ob1 = ob + tr.off1;
        const long md = mask ? mask - reinterpret_cast<const bf16_t*>(a.out) : 0;
        const int stride = (int)tr.stride;              // (< 2^31 / BM: map pixels <= 1024 x channels <= 2048, or channels alone)
#pragma unroll
        for (int h = 0; h < NR; h += HALF) {
            uint4 rowv[HALF];
            int eo[HALF];
            unsigned kbits = 0u;
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int r = wave * RW + (h + j) * RPI + lane / CPR;
                rowv[j] = *reinterpret_cast<const uint4*>(lds + r * ROWB + (lane & (CPR - 1)) * 16);
                const int c = (lane & (CPR - 1)) ^ (r & (CPR - 1));
                const bool k = r >= tr.rs1;
                const int idx = k ? r - tr.rs1 : r;
                eo[j] = idx < (k ? tr.nv1 : tr.nv0) ? idx * stride + c * 8 : -1;
                kbits |= k ? (1u << j) : 0u;
            }
            if (mask) {
                uint4 mk[HALF];
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    mk[j] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                    if (eo[j] >= 0) mk[j] = *reinterpret_cast<const uint4*>(((kbits >> j) & 1u ? ob1 : ob0) + eo[j] + md);
                }
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    rowv[j].x = sel(rowv[j].x, mk[j].x); rowv[j].y = sel(rowv[j].y, mk[j].y);
                    rowv[j].z = sel(rowv[j].z, mk[j].z); rowv[j].w = sel(rowv[j].w, mk[j].w);
                }
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j)
                if (eo[j] >= 0) {
                    // non-temporal: a 128 KiB output tile is not read again by this CU, and the maps these launches write (0.2-0.8 GB)
                    // outlast the L2 whatever reads them next -- no write-allocate in front of the operand tiles the K loops re-read
                    // (same box, 9 interleaved runs: -0.1 .. -0.2 ms per step; tools/ab_libs.sh)
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_nt;
                    const u32x4_nt vv = {rowv[j].x, rowv[j].y, rowv[j].z, rowv[j].w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_nt*>(const_cast<bf16_t*>(((kbits >> j) & 1u ? ob1 : ob0) + eo[j])));
                }
        }
    } else {
        // strided / offset outputs (parity-class launches of strided data gradients, the mask head's 2x2 deconvolution): rows are
        // linear only within one map row, so every row is decoded -- with the decode's scalars held in registers (LOFT_KEEP_S:
        // the loop otherwise re-fetches a dozen kernarg fields per row)
        int aM = a.M, aB = a.B, aOW = a.OW, aOHf = a.OHf, aOWf = a.OWf, aos = a.os, aoy = ooy, aox = oox, aCout = a.Cout;
        unsigned m1 = a.ohw_mul, s1 = a.ohw_sh, m2 = a.ow_mul, s2 = a.ow_sh;
        LOFT_KEEP_S(aM); LOFT_KEEP_S(aB); LOFT_KEEP_S(aOW); LOFT_KEEP_S(aOHf); LOFT_KEEP_S(aOWf); LOFT_KEEP_S(aos); LOFT_KEEP_S(aoy);
        LOFT_KEEP_S(aox); LOFT_KEEP_S(aCout); LOFT_KEEP_S(m1); LOFT_KEEP_S(s1); LOFT_KEEP_S(m2); LOFT_KEEP_S(s2);
        const bool pmaj = a.pixmajor != 0;
#pragma unroll 1
        for (int it = 0; it < RW / RPI; ++it) {
            const int r = wave * RW + it * RPI + lane / CPR;
            const int c = (lane & (CPR - 1)) ^ (r & (CPR - 1));
            const int m = m0 + r;
            const bf16_t* rp = nullptr;
            if (pmaj) {
                if (m < aM) rp = pipe_row_ptr(a, reinterpret_cast<const bf16_t*>(a.out), out_g, m, n0, ohw, ooy, oox);
            } else if (m < aM) {
                const int b = fastdiv(m, m1, s1), rem = m - b * ohw;
                const int oy = fastdiv(rem, m2, s2), ox = rem - oy * aOW;
                if (b < aB) rp = ob + (((long)b * aOHf + oy * aos + aoy) * aOWf + ox * aos + aox) * aCout;
            }
            if (rp) {
                uint4 v = *reinterpret_cast<const uint4*>(lds + r * ROWB + (lane & (CPR - 1)) * 16);
                const bf16_t* p = rp + c * 8;
                if (mask) {
                    const uint4 mk = *reinterpret_cast<const uint4*>(mask + (p - reinterpret_cast<const bf16_t*>(a.out)));
                    v.x = sel(v.x, mk.x); v.y = sel(v.y, mk.y); v.z = sel(v.z, mk.z); v.w = sel(v.w, mk.w);
                }
                *reinterpret_cast<uint4*>(const_cast<bf16_t*>(p)) = v;
            }
        }
    }
    kstamp(48);
}

// Epilogue of the operand-plane launches (PL instances; loft_conv_tap_planes): fp32 everywhere -- the accumulators hold the sum of
// the launch's plane products, scaled back by 1 / (scale_x * scale_w) when the planes are scaled; bias, residual, ReLU and the
// ReLU-backward mask as in conv_epilogue, on float tensors; every lane stores its four consecutive couts of a pixel as one
// 16-byte access.  (The direct form: the K loop of a plane launch is 3 or 6 terms long, the epilogue a few percent of the tile.)
template <int MJ, int NW>
__device__ __forceinline__ void pipe_epilogue_f32(const ConvArgs& a, const TileRows& tr, f32x16 (&acc)[2][4], int g, int m0, int n0,
                                                  int wave, int lane, int ohw) {
    const int wm = wave >> 2, wn = wave & 3, frow = lane & 31, fq = lane >> 5;
    const long out_g = (long)g * a.out_gs;
    const float sc = a.amax_x ? planes_scale_of(*a.amax_x, true) * planes_scale_of(*a.amax_w, true) : 1.f;
    const float* bias = a.bias ? a.bias + (long)g * a.bias_gs : nullptr;
    const float* res = reinterpret_cast<const float*>(a.residual);
    const float* msk = reinterpret_cast<const float*>(a.mask);
    float* out = reinterpret_cast<float*>(a.out);
    const bool relu = a.relu != 0;
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int r = wm * (32 * MJ) + j * 32 + frow;
        long roff;
        bool ok;
        if (tr.lin) roff = pipe_row_off(tr, r, ok);
        else {
            const int m = m0 + r;
            ok = m < a.M;
            roff = 0;
            if (ok) {
                int b, rem;
                pipe_row_decode(a, m, ohw, b, rem);
                ok = b < a.B;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                roff = (((long)b * a.OHf + oy * a.os + a.oo_y) * a.OWf + ox * a.os + a.oo_x) * a.Cout;
            }
        }
        if (!ok) continue;
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * (32 * NW) + i * 32 + 8 * gq + 4 * fq;
                const long o = out_g + roff + n;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][gq * 4 + e] * sc;
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (res) {
                    const float4 rv = *reinterpret_cast<const float4*>(res + o);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (msk) {
                    const float4 mv = *reinterpret_cast<const float4*>(msk + o);
                    v[0] = mv.x > 0.f ? v[0] : 0.f; v[1] = mv.y > 0.f ? v[1] : 0.f;
                    v[2] = mv.z > 0.f ? v[2] : 0.f; v[3] = mv.w > 0.f ? v[3] : 0.f;
                }
                *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if ((v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3])) vmax = __builtin_inff();
            }
    }
    if (a.amax_out) {           // one fire-and-forget atomic per wave (non-negative floats order as their bits)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax_out), __float_as_uint(vmax));
    }
}

// The same fp32 epilogue with the tile collected in LDS (round 6; the 256 x 256 plane launches): the direct form above stores 32 bytes
// per output row and instruction (two lanes per row, 32 rows per wave-level store) and reads residual / mask tiles the same way;
// here the accumulators go to LDS -- scaled, bias added: the fp32 values the direct form has at that point -- and leave as whole
// 1 KiB rows (64 lanes x 16 bytes), residual and mask rows arriving with the same coalesced pattern.  A 256 x 256 fp32 tile is
// 256 KiB, the array 128 KiB: two passes of 128 rows (pass h = the four waves with wm == h, whose wave tiles are exactly those
// rows).  16-byte chunk c of local row lr sits at position c ^ (lr & 63) (64 chunks per row; conflict-free for the 8-lane groups
// of ds_write_b128: eight consecutive rows, one chunk each).  Same operations in the same order: bit-identical to the direct form
// (tests/test_planes_gpu.py::test_staged_fp32_epilogue_is_bit_identical_to_the_direct_one).
__device__ __forceinline__ void pipe_epilogue_f32_staged(const ConvArgs& a, const TileRows& tr, f32x16 (&acc)[2][4], char* lds, int g, int n0,
                                                         int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3, frow = lane & 31, fq = lane >> 5;
    const long out_g = (long)g * a.out_gs;
    const float sc = a.amax_x ? planes_scale_of(*a.amax_x, true) * planes_scale_of(*a.amax_w, true) : 1.f;
    const float* bias = a.bias ? a.bias + (long)g * a.bias_gs + n0 : nullptr;
    const float* res = reinterpret_cast<const float*>(a.residual);
    const float* msk = reinterpret_cast<const float*>(a.mask);
    float* out = reinterpret_cast<float*>(a.out);
    const bool relu = a.relu != 0;
    float vmax = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave is done with the K loop's fragments
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c = wn * 16 + i * 8 + 2 * gq + fq;        // logical 16-byte chunk = couts [4 c, 4 c + 4) of the tile
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int lr = j * 32 + frow;
                        float4 v = make_float4(acc[i][j][gq * 4 + 0] * sc, acc[i][j][gq * 4 + 1] * sc, acc[i][j][gq * 4 + 2] * sc,
                                               acc[i][j][gq * 4 + 3] * sc);
                        if (bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                        *reinterpret_cast<float4*>(lds + lr * 1024 + ((c ^ (lr & 63)) << 4)) = v;
                    }
                }
        }
        __syncthreads();
        // copy-out: wave w serves local rows [16 w, 16 w + 16), four at a time (their residual / mask rows requested together)
#pragma unroll 1
        for (int it = 0; it < 16; it += 4) {
            float4 v[4], rv[4], mv[4];
            long o[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int lr = wave * 16 + it + u;
                const long roff = pipe_row_off(tr, h * 128 + lr, ok[u]);
                o[u] = out_g + roff + n0 + ((lane ^ (lr & 63)) << 2);
                v[u] = *reinterpret_cast<const float4*>(lds + lr * 1024 + lane * 16);
                rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                mv[u] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (ok[u] && res) rv[u] = *reinterpret_cast<const float4*>(res + o[u]);
                if (ok[u] && msk) mv[u] = *reinterpret_cast<const float4*>(msk + o[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                float w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                if (res) { w[0] += rv[u].x; w[1] += rv[u].y; w[2] += rv[u].z; w[3] += rv[u].w; }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = fmaxf(w[e], 0.f);
                }
                if (msk) {
                    w[0] = mv[u].x > 0.f ? w[0] : 0.f; w[1] = mv[u].y > 0.f ? w[1] : 0.f;
                    w[2] = mv[u].z > 0.f ? w[2] : 0.f; w[3] = mv[u].w > 0.f ? w[3] : 0.f;
                }
                *reinterpret_cast<float4*>(out + o[u]) = make_float4(w[0], w[1], w[2], w[3]);
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3]))));
                if ((w[0] != w[0]) | (w[1] != w[1]) | (w[2] != w[2]) | (w[3] != w[3])) vmax = __builtin_inff();
            }
        }
        __syncthreads();                               // (the next pass overwrites the rows)
    }
    if (a.amax_out) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o2));
        if (lane == 0 && vmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax_out), __float_as_uint(vmax));
    }
}

// VAR bits (A/B experiments, selected through the C-ABI variant argument; 0 = the shipped schedule):
//   1 TRACE      lane 0 of every wave stores s_memtime at every barrier exit of K-tiles 8..11 to a.trace ([wave][32] u64)
//   2 NOPRIO     no s_setprio around the MFMA clusters
//   4 OLDORDER   fragment reads 12 / 4 / 8 / 0 per phase (W c0 read in L0 instead of the previous tile's L3)
// MODE 0: "phase" schedule above.  MODE 1: "stream" schedule (below, after the phase loop's description): every wave runs ONE
// software-pipelined instruction stream with a single barrier per K-tile.
// MJ: 32-pixel blocks per wave.  4: the 256-pixel tile.  2 / 1 (stream schedule only): a 128- / 64-pixel x 256-cout tile for launches whose
// 256-pixel tiles cannot fill the chip (layer3's 64 x 64 maps: 128 tiles) -- same staging, swizzle, schedule and epilogue, wave
// tile 64 x 64; LDS reads and copies per FLOP rise by a third, twice as many workgroups.
// NW: 32-cout blocks per wave.  2: 256 couts per tile.  1 (stream schedule only): 128 couts per tile (layer2's 128-channel convs: Cout
// is not a multiple of 256) -- one weight fragment per sub-step, 256-byte output rows in the staged epilogue.
// PL: operand-plane launch (ConvArgs block 4): the K loop runs nterms x taps, the epilogue is pipe_epilogue_f32 (stream schedule only).
// R32 ("ring32", round 5; schedule below): the 256 x 256 tile with 32-CHANNEL K-tiles on a FOUR-stage ring (64-byte LDS rows).
// XF ("activations first", round 6; LOFT_CONV_XFIRST): the two-stage stream schedule with the roles of the two operands' copies swapped --
// X(t+2) is requested right behind SYNC(t) (sub-step ks3), W(t+1) in ks0 -- see the stream schedule's description below.
// LEAN (round 6; LOFT_CONV_LEAN / LOFT_CONV_LEANX): the same schedule with fewer INSTRUCTIONS per K-tile.  What the counters say
// about the copies' cost (profiles/round6_probes/pmc_stream_foa.txt): TA 36 % busy, TCP -> L2 read latency 136 cycles on
// average, no address / command FIFO ever full, SQ_WAIT_ANY identical with and without copies -- the memory path is not what
// the K loop waits for.  What the copy-free variant lacks is the copies' ISSUE work: per K-tile and wave ~31 vector and ~58
// scalar instructions (SQ_ACTIVE_INST_SCA x 2.8, SQ_ACTIVE_INST_VALU x 1.5) that an in-order wave issues between its own MFMAs.
// LEAN: (1) the K-tile sequence is chunk-major at compile time (the runtime select between the two orders computed both every
// step); (2) tap offset + chunk offset are ONE 64-bit scalar per K-tile (hipcc re-associated the sum into two 64-bit vector
// adds per copy); (3) tiles whose rows are all valid for every tap they run -- every FOA / mask tile: one segment of 256 RoIs
// at one position -- issue their activation copies without the zero-page select (4 vector instructions per copy).
template <int MODE, int VAR, int MJ = 4, int NW = 2, bool PL = false, bool R32 = false, bool XF = false, bool LEAN = false>
__global__ __launch_bounds__(512) void conv_tap_pipe_kernel(const ConvArgs a) {
    static_assert(!LEAN || (MODE == 1 && VAR == 0 && MJ == 4 && NW == 2 && !R32), "LEAN exists for the two-stage 256 x 256 stream schedule (16-bit and operand-plane launches)");
    static_assert(MODE <= 2, "MODE 0 phase, 1 stream, 2 role-split stream");
    static_assert(!XF || (MODE == 1 && VAR == 0 && MJ == 4 && NW == 2 && !PL && !R32), "activations-first exists for the two-stage 256 x 256 stream schedule");
    static_assert(!PL || (MODE == 1 && VAR == 0), "operand planes exist for the stream schedule");
    static_assert(!R32 || (MODE == 1 && VAR == 0 && MJ == 4 && NW == 2), "the 32-channel ring exists for the stream schedule's 256 x 256 tile");
    static_assert(MJ == 4 || ((MJ == 2 || MJ == 1) && MODE == 1 && !(VAR & 2)), "the 128- / 64-pixel tiles exist for the stream schedule");
    static_assert(NW == 2 || (NW == 1 && MODE == 1 && !(VAR & 2)), "the 128-cout tile exists for the stream schedule");
    constexpr int BN = 128 * NW;
    constexpr int NI = R32 ? 2 : MJ;                                // activation rows staged per thread (64 rows apart; R32: two, 16 apart)
    constexpr int BM = 64 * MJ;
    constexpr bool ABL = VAR & 8;                                   // timing ablations (results are WRONG): sub-code in bits 1-2
    constexpr bool TRACE = VAR & 1, NOPRIO = !ABL && (VAR & 2), OLDORDER = !ABL && (VAR & 4);
    constexpr bool NOGLDS = ABL && ((VAR >> 1) & 3) == 0, NOREADS = ABL && ((VAR >> 1) & 3) == 1, NOMFMA = ABL && ((VAR >> 1) & 3) == 2,
                   NOSEL = false;                      // (sub-code 3 used to skip the zero-page select: faults on padded maps; retired)
    // VAR 14 (round 3): the copy volume of a SHARED HALO PATCH, timing only -- the activation copies of a 64-channel chunk are
    // issued for its first tap alone (one tile's worth of rows, where a patch would stage 18 x 18 = 1.27 tiles), the weight copies
    // for every tap as before.  An upper bound on what any patch scheme can return.
    constexpr bool HALOX = ABL && ((VAR >> 1) & 3) == 3;
    // VAR 4 in the stream schedule (round 3, timing only): per K-tile only ONE wave of every SIMD issues global->LDS copies -- waves
    // 0-3 in even tiles, 4-7 in odd ones -- and it issues twice as many (its own pieces twice: same instruction count and bytes per CU)
    constexpr bool ALTCOPY = MODE == 1 && VAR == 4;
    // Stage ring.  The 256-pixel tile double-buffers (4 x 32 KiB).  The 128- / 64-pixel tiles (layer3 / layer4: few workgroups, one per
    // CU, K-tiles of only 256-512 MFMA cycles) take THREE stages: a K-tile's copies are requested two tiles ahead and retired
    // with a counted vmcnt, because an L2 round trip is longer than one of their K-tiles (two stages left them latency-bound at
    // 20-35 % of the MFMA rate).  Weights 3 x NW x 16 KiB, activations 3 x MJ x 8 KiB behind them; also the 128-cout tile (NW = 1).
    constexpr int NS = (MODE == 1 && VAR == 0 && !(MJ == 4 && NW == 2)) ? 3 : 2;
    // MODE 2 ("roles", round 4; see the schedule below): THREE weight stages + two activation stages = all 160 KiB of the CU
    constexpr bool ROLES = MODE == 2;
    static_assert(!ROLES || (MJ == 4 && NW == 2 && VAR == 0), "the role-split schedule exists for the 256 x 256 tile");
    // (R32: four stages of [256 rows x 64 B] per operand: weights 4 x 16 KiB, activations 4 x 16 KiB behind them)
    constexpr int PBW = R32 ? 16384 : (ROLES ? PBUF : (NS == 3 ? NW * 16384 : PBUF)), PBX = R32 ? 16384 : (ROLES ? PBUF : (NS == 3 ? MJ * 8192 : PBUF));
    constexpr int PXO = R32 ? 4 * 16384 : (ROLES ? 3 * PBUF : (NS == 3 ? 3 * PBW : PX_OFF));
    constexpr int LDSZ = R32 ? PLDS : (ROLES ? 5 * PBUF : (NS == 3 ? 3 * PBW + 3 * PBX : PLDS));
    // rows a thread holds staging pointers for: its own four (i = 0..3 -> tile row i*64 + wave*8 + lrow) and, in the role-split
    // schedule, the four of its SIMD partner (wave ^ 4), whose activation pieces the copier wave issues as well
    constexpr int NI2 = ROLES ? 2 * NI : NI;
    __shared__ __attribute__((aligned(16))) char lds[LDSZ];
    unsigned long long kst0 = 0ull, kst1 = 0ull, kst2 = 0ull;
    const int trace_b0 = gridDim.x > 1100 ? 1024 : 0;        // TRACE: a workgroup of a later round (steady state) when there is one
    auto kstamp = [&](int slot) {                             // TRACE, outside the K loop: time stamp straight to the buffer
        if constexpr (TRACE && MODE == 1) {
            if ((int)blockIdx.x >= trace_b0 && (int)blockIdx.x < trace_b0 + 2 && blockIdx.y == 0 && blockIdx.z == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if ((threadIdx.x & 63) == 0)
                    (reinterpret_cast<unsigned long long*>(a.trace) + ((long)(blockIdx.x - trace_b0) * 8 + (threadIdx.x >> 6)) * 64)[slot] = now;
            }
        }
    };
    if constexpr (TRACE && MODE == 1) kst0 = __builtin_amdgcn_s_memtime();
    // every scalar of the set-up requested NOW, in a handful of wide loads behind one wait (ConvArgs block 1)
    // (ONE statement per ~20 operands: separate asm statements are ordered among themselves and each would wait for its own load)
    asm volatile("" :: "s"(a.src), "s"(a.wgt), "s"(a.zero_page), "s"(a.out), "s"(a.src_gs), "s"(a.wgt_gs), "s"(a.gxy_mul), "s"(a.gxy_sh),
                 "s"(a.gx_mul), "s"(a.gx_sh), "s"(a.gy_mul), "s"(a.gy_sh), "s"(a.nfast), "s"(a.pixmajor), "s"(a.pointwise), "s"(a.T),
                 "s"(a.B), "s"(a.IH), "s"(a.IW), "s"(a.Cin), "s"(a.Cout), "s"(a.OH), "s"(a.OW), "s"(a.M), "s"(a.ss), "s"(a.pm_S),
                 "s"(a.pm_P), "s"(a.pms_mul), "s"(a.pms_sh));
    asm volatile("" :: "s"(a.pmp_mul), "s"(a.pmp_sh), "s"(a.ohw_mul), "s"(a.ohw_sh), "s"(a.ow_mul), "s"(a.ow_sh), "s"(a.tap_major), "s"(a.krot),
                 "s"(a.dy_pk), "s"(a.dx_pk), "s"(a.wt_pk), "s"(a.pk_ok));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    // (host-computed multiply-high constants: five runtime integer divisions at kernel entry were ~1k cycles per workgroup)
    const int bz = fastdiv(V, a.gxy_mul, a.gxy_sh), Vg = V - bz * (int)(gridDim.x * gridDim.y);
    int bx, by;
    if (a.nfast) { bx = fastdiv(Vg, a.gy_mul, a.gy_sh); by = Vg - bx * (int)gridDim.y; }
    else { by = fastdiv(Vg, a.gx_mul, a.gx_sh); bx = Vg - by * (int)gridDim.x; }
    const int m0 = bx * BM, n0 = by * BN;
    const int g = bz;
    const bf16_t* src = a.src + (long)g * a.src_gs;
    const bf16_t* wgt = a.wgt + (long)g * a.wgt_gs;
    const int lrow = lane >> 3, lchunk = lane & 7;
    const int ohw = a.OH * a.OW;
    const int kchunks = a.Cin / BK;

    // ---- rows this thread stages: i = 0..3 -> tile row i*64 + wave*8 + lrow (rows 64 apart share their swizzle)
    // R32: a 1 KiB piece = 16 rows x 64 B (lane -> row lane >> 2, 16-byte position lane & 3); a wave stages rows [32 wave, +32) of
    // both operands as two pieces; logical 16-byte chunk q of row r sits at position q ^ ((r >> 2) & 3) (swizzle on the SOURCE:
    // the thread at position p fetches chunk p ^ ((r >> 2) & 3); rows 16 / 32 apart share it), conflict-free for the
    // ds_read_b128 lane groups of the fragment reads (16 rows x one chunk each = the 16 slots of four 256-byte bank rows).
    const int srow = R32 ? wave * 32 + (lane >> 2) : wave * 8 + lrow;
    const int schk = R32 ? (((lane & 3) ^ ((srow >> 2) & 3)) * 8) : swz(srow, lchunk) * 8;     // element offset of this thread's chunk
    const int b_off0 = (n0 + srow) * a.Cin + schk;                        // weight row of piece-row i: + i * b_step
    const int b_step = (R32 ? 16 : 64) * a.Cin;
    // Per-tap tables live in LANES (lane t = tap t) and are fetched with v_readlane: a kernarg (SMEM) load indexed by a loop
    // counter costs a ~200-cycle round trip each (the tap-mask loop of round 1's kernels: 36 of them, 7k cycles per workgroup),
    // and inside the K loop it would make hipcc wait lgkmcnt(0), i.e. for every fragment read in flight.
    int tab_dy = 0, tab_dx = 0, tab_a = 0, tab_w = 0;
    if (a.pointwise) tab_w = a.wt[0] * a.Cout * a.Cin;          // (1x1 / FC: no per-lane table loads; offsets are zero)
    else if (a.pk_ok) {
        const int sh = (lane & 15) * 4;
        tab_dy = (int)((a.dy_pk >> sh) & 15ull) - 8; tab_dx = (int)((a.dx_pk >> sh) & 15ull) - 8;
        tab_a = (tab_dy * a.IW + tab_dx) * a.Cin;
        tab_w = (int)((a.wt_pk >> sh) & 15ull) * a.Cout * a.Cin;
    } else if (lane < a.T) {
        tab_dy = a.dy[lane]; tab_dx = a.dx[lane];
        tab_a = (tab_dy * a.IW + tab_dx) * a.Cin;
        tab_w = a.wt[lane] * a.Cout * a.Cin;
    }
    kstamp(40);
    const bf16_t* a_ptr[NI2];
    unsigned a_mask[NI2];
    int a_iy[NI2], a_ix[NI2];
    auto tile_row = [&](int i) { return R32 ? wave * 32 + i * 16 + (lane >> 2) : (i % NI) * 64 + (i < NI ? wave : (wave ^ 4)) * 8 + lrow; };
    // element offset of the thread's 16-byte chunk inside the K-tile's channel range, for tile row `row`
    auto row_chunk = [&](int row) { return R32 ? (((lane & 3) ^ ((row >> 2) & 3)) * 8) : swz(row, lchunk) * 8; };
    unsigned tmask = 0xffffffffu;       // taps any row of the tile needs (pixel-major tiles skip the others)
    if (a.pixmajor) {
        // pixel-major rows (RoI maps): the tile's rows are linear in the row index within each of its (at most two) segments,
        // and the tap masks are functions of the segment's pixel position only -- everything but one select + multiply-add per
        // staged row runs on the scalar unit (TileRows above; pix_ok guarantees ss == os == 1)
        int rs1, b00, b10, pos0, pos1;
        pipe_pm_tile(a, m0, rs1, b00, b10, pos0, pos1);
        const int oy0 = fastdiv(pos0, a.ow_mul, a.ow_sh), ox0 = pos0 - oy0 * a.OW;
        const int oy1 = fastdiv(pos1, a.ow_mul, a.ow_sh), ox1 = pos1 - oy1 * a.OW;
        unsigned mk0 = 0u, mk1 = 0u;
        for (int t = 0; t < a.T; ++t) {
            const int dy = __builtin_amdgcn_readlane(tab_dy, t), dx = __builtin_amdgcn_readlane(tab_dx, t);
            const int iy0_ = oy0 * a.ss + dy, ix0_ = ox0 * a.ss + dx, iy1_ = oy1 * a.ss + dy, ix1_ = ox1 * a.ss + dx;
            mk0 |= ((iy0_ >= 0) & (iy0_ < a.IH) & (ix0_ >= 0) & (ix0_ < a.IW)) ? (1u << t) : 0u;
            mk1 |= ((iy1_ >= 0) & (iy1_ < a.IH) & (ix1_ >= 0) & (ix1_ < a.IW)) ? (1u << t) : 0u;
        }
        const int rows = pipe_clamp(a.M - m0, 0, BM);
        const int nv0 = pipe_clamp(a.B - b00, 0, rs1 < rows ? rs1 : rows);
        const int nv1 = pipe_clamp(a.B - b10, 0, rows - rs1 > 0 ? rows - rs1 : 0);
        const long in0 = ((long)(b00 * a.IH + oy0 * a.ss) * a.IW + ox0 * a.ss) * a.Cin;
        const long in1 = ((long)(b10 * a.IH + oy1 * a.ss) * a.IW + ox1 * a.ss) * a.Cin;
        const long istr = (long)a.IH * a.IW * a.Cin;
        tmask = (nv0 > 0 ? mk0 : 0u) | (nv1 > 0 ? mk1 : 0u);
#pragma unroll
        for (int i = 0; i < NI2; ++i) {
            const int row = tile_row(i);
            const bool k = row >= rs1;
            const int idx = k ? row - rs1 : row;
            const bool ok = idx < (k ? nv1 : nv0);
            a_ptr[i] = ok ? src + ((k ? in1 : in0) + (long)idx * istr + row_chunk(row)) : src;
            a_mask[i] = ok ? (k ? mk1 : mk0) : 0u;
            a_iy[i] = 0; a_ix[i] = 0;
        }
    } else {
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
        const int row = tile_row(i);
        const int m = m0 + row;
        a_ptr[i] = src;
        a_mask[i] = 0u;
        a_iy[i] = -(1 << 20); a_ix[i] = -(1 << 20);
        if (a.pointwise) {                       // input pixel index = m: no decode, the one tap is always inside the map
            if (m < a.M) {
                a_iy[i] = 0; a_ix[i] = 0;
                a_mask[i] = 1u;
                a_ptr[i] = src + ((long)m * a.Cin + row_chunk(row));
            }
            continue;
        }
        if (m < a.M) {
            const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
            const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
            a_iy[i] = oy * a.ss; a_ix[i] = ox * a.ss;
            a_ptr[i] = src + ((long)(b * a.IH * a.IW + a_iy[i] * a.IW + a_ix[i]) * a.Cin + row_chunk(row));
        }
    }
    kstamp(41);
    if (!a.pointwise) {
        for (int t = 0; t < a.T; ++t) {
            const int dy = __builtin_amdgcn_readlane(tab_dy, t), dx = __builtin_amdgcn_readlane(tab_dx, t);
#pragma unroll
            for (int i = 0; i < NI2; ++i) {
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                a_mask[i] |= ((iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW)) ? (1u << t) : 0u;
            }
        }
    }
    }
    kstamp(42);
    int nk = a.T * kchunks;
    if (a.pixmajor) nk = __popc(tmask) * kchunks;
    // LEAN (3): every staged row of this workgroup valid for every tap the tile runs?  (wave-uniform over the whole workgroup:
    // a per-wave answer would be enough for correctness, the branch below only has to be wave-uniform)
    bool all_valid = false;
    if constexpr (LEAN) {
        const unsigned need = a.pixmajor ? tmask : (a.T >= 32 ? 0xffffffffu : ((1u << a.T) - 1u));
        bool mine = true;
#pragma unroll
        for (int i = 0; i < NI; ++i) mine = mine && ((a_mask[i] & need) == need);
        all_valid = __builtin_amdgcn_readfirstlane((int)(__ballot(mine) == ~0ull)) != 0;
    }
    // ---- wave-uniform state of the K-tile being STAGED: tap, channel offset, the tap's source / weight offsets
    int st_t = 0, st_c = a.krot ? (V % kchunks) * BK : 0;        // (krot: this workgroup's first channel chunk)
    while (st_t < a.T - 1 && !((tmask >> st_t) & 1u)) ++st_t;
    long st_aoff = (long)__builtin_amdgcn_readlane(tab_a, st_t);
    const bf16_t* st_w = wgt + (long)__builtin_amdgcn_readlane(tab_w, st_t);
    // (stream mode: the weight pieces are issued one K-tile ahead of the activation pieces -> their own tap / channel state)
    int sw_t = st_t, sw_c = st_c;
    // K order.  Chunk-major (default): for every 64-channel chunk, all taps.  The three dx taps of an input row -- and the rows a
    // tile shares with its neighbours -- then re-read the same 128-byte segments within a few K-tiles, while they are still in
    // the XCD's L2; in tap-major order (a.tap_major: the lock-step kernels' order, bit-identical sums) a re-read comes kchunks
    // K-tiles later, after the 32 CUs of the XCD have streamed ~4 MB through that L2: measured 5.2x the input bytes fetched
    // from HBM per 3x3 launch (tools/pmc_traffic_shapes.sh).
    const int first_t = st_t;
    auto next_tap = [&](int t) {
        ++t;
        while (t < a.T && !((tmask >> t) & 1u)) ++t;
        return t;
    };
    // The K-tile sequence of the stream schedules WITHOUT control flow (round 4).  The ISA of the loop body showed the tap / chunk
    // bookkeeping -- "next tap that some row of this tile needs", wrap to the next channel chunk, table look-ups -- as ~100 scalar
    // instructions with two nested search loops and a dozen taken branches, sitting between the MFMA pairs of the two copy
    // sub-steps of EVERY K-tile, in both waves of a SIMD at the same moment.  Here the taps the tile needs are compacted ONCE into
    // lanes 0..nv-1 (vt_a / vt_w: activation / weight element offsets, vt_m: the tap's bit in the row masks) and a step is an
    // increment, a compare and four s_cselect.
    int vt_a = 0, vt_w = 0, vt_m = 0, nv = 0;
    if constexpr (PL) {
        // operand planes: the list holds every needed tap once per TERM, with the term's plane offsets added -- the K loop below
        // then walks (chunk, term, tap) and all terms meet in the accumulators; vt_m stays the SPATIAL tap (the row masks' bit)
        for (int p = 0; p < a.nterms; ++p) {
            const int xo = a.xoff[p], wo = a.woff[p];
            if (a.pointwise) {
                vt_a = lane == nv ? xo : vt_a;
                vt_w = lane == nv ? __builtin_amdgcn_readfirstlane(tab_w) + wo : vt_w;
                ++nv;
            } else {
                for (int t = 0; t < a.T; ++t) {
                    if ((tmask >> t) & 1u) {
                        const int ta = __builtin_amdgcn_readlane(tab_a, t), tw = __builtin_amdgcn_readlane(tab_w, t);
                        vt_a = lane == nv ? ta + xo : vt_a;
                        vt_w = lane == nv ? tw + wo : vt_w;
                        vt_m = lane == nv ? t : vt_m;
                        ++nv;
                    }
                }
            }
        }
        nk = nv * kchunks;
        st_t = __builtin_amdgcn_readlane(vt_m, 0);
        st_aoff = (long)__builtin_amdgcn_readlane(vt_a, 0);
        st_w = wgt + (long)__builtin_amdgcn_readlane(vt_w, 0);
        sw_t = st_t;
    } else {
    for (int t = 0; t < a.T; ++t) {
        if ((tmask >> t) & 1u) {
            const int ta = __builtin_amdgcn_readlane(tab_a, t), tw = __builtin_amdgcn_readlane(tab_w, t);
            vt_a = lane == nv ? ta : vt_a;
            vt_w = lane == nv ? tw : vt_w;
            vt_m = lane == nv ? t : vt_m;
            ++nv;
        }
    }
    if (a.pointwise) { vt_w = tab_w; nv = 1; }                  // (tab_w is uniform there, tab_a / the mask bit are 0)
    }
    int xj = 0, wj = 0;                                          // positions in the compact list of the tiles being staged
    auto seq_step = [&](int& j, int& c) {
        if constexpr (LEAN) {           // chunk-major only (the launcher routes tap-major requests elsewhere)
            const int jn = j + 1;
            const bool wj_ = jn >= nv;
            const int cn = c + BK;
            j = wj_ ? 0 : jn;
            c = wj_ ? (cn == a.Cin ? 0 : cn) : c;
            return;
        }
        const int jn = j + 1, cn = c + BK;
        const bool wj_ = jn >= nv, wc_ = cn == a.Cin;
        // chunk-major: next tap, wrapping to the next chunk (which itself wraps under krot); tap-major: next chunk, wrapping to the next tap
        const int j_cm = wj_ ? 0 : jn, c_cm = wj_ ? (wc_ ? 0 : cn) : c;
        const int j_tm = wc_ ? jn : j, c_tm = wc_ ? 0 : cn;
        j = a.tap_major ? j_tm : j_cm;
        c = a.tap_major ? c_tm : c_cm;
    };
    auto advance_w = [&]() {
        seq_step(wj, sw_c);
        sw_t = __builtin_amdgcn_readlane(vt_m, wj);
        st_w = wgt + (long)__builtin_amdgcn_readlane(vt_w, wj);
    };
    auto advance_x = [&]() {
        seq_step(xj, st_c);
        st_t = __builtin_amdgcn_readlane(vt_m, xj);
        st_aoff = (long)__builtin_amdgcn_readlane(vt_a, xj);
    };
    auto advance = [&]() {
        if (a.tap_major) {
            st_c += BK;
            if (st_c == a.Cin) {
                st_c = 0;
                st_t = next_tap(st_t);
                if (st_t < a.T) {
                    st_aoff = (long)__builtin_amdgcn_readlane(tab_a, st_t);
                    st_w = wgt + (long)__builtin_amdgcn_readlane(tab_w, st_t);
                }
            }
        } else {
            st_t = next_tap(st_t);
            if (st_t >= a.T) { st_t = first_t; st_c += BK; if (st_c == a.Cin) st_c = 0; }
            st_aoff = (long)__builtin_amdgcn_readlane(tab_a, st_t);
            st_w = wgt + (long)__builtin_amdgcn_readlane(tab_w, st_t);
        }
    };
    bool in_loop = false;
    auto issue_w = [&](auto halfc, auto bufc) {
        constexpr int H = decltype(halfc)::value, B = decltype(bufc)::value;
        if constexpr (NOGLDS) { if (in_loop) return; }
        if constexpr (2 * H >= 2 * NW) return;                    // (128-cout tile: the second half does not exist)
        const bf16_t* wt = st_w + (MODE >= 1 ? sw_c : st_c) + b_off0;
#pragma unroll
        for (int i = 2 * H; i < 2 * H + 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wt + (long)i * b_step),
                                             (lds_ptr_t)(lds + PW_OFF + B * PBW + (i * 64 + wave * 8) * 128), 16, 0, 0);
    };
    auto issue_x = [&](auto halfc, auto bufc, auto avc) {        // avc: std::true_type = every row valid for every tap (LEAN)
        constexpr int H = decltype(halfc)::value, B = decltype(bufc)::value;
        constexpr bool AV = decltype(avc)::value;
        if constexpr (NOGLDS) { if (in_loop) return; }
        if constexpr (HALOX) { if (in_loop && st_t != first_t) return; }
        if constexpr (2 * H >= NI) return;                        // (128- / 64-pixel tiles: the second half does not exist)
        long aoff = st_aoff + st_c;
        if constexpr (LEAN) {
            asm volatile("" : "+s"(aoff));                        // ONE scalar 64-bit sum (hipcc otherwise adds the two parts per lane)
            if constexpr (AV) {
#pragma unroll
                for (int i = 2 * H; i < 2 * H + 2; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a_ptr[i] + aoff), (lds_ptr_t)(lds + PXO + B * PBX + (i * 64 + wave * 8) * 128), 16, 0, 0);
                return;
            }
        }
#pragma unroll
        for (int i = 2 * H; i < (2 * H + 2 < NI ? 2 * H + 2 : NI); ++i) {
            const bf16_t* p = (NOSEL || ((a_mask[i] >> st_t) & 1u)) ? a_ptr[i] + aoff : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(lds + PXO + B * PBX + (i * 64 + wave * 8) * 128), 16, 0, 0);
        }
    };
    using c0_t = std::integral_constant<int, 0>;
    using c1_t = std::integral_constant<int, 1>;

    // Accumulator zeroing (128 v_mov per wave: ~1k cycles of a SIMD's VALU for its two waves) and the fragment base addresses are
    // issued AFTER the first K-tile's copies have been requested (late_init below): they run under that fetch's latency instead
    // of in front of it.
    f32x16 acc[2][4];
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fq = lane >> 5;
    // fragment bases per 16-channel sub-step ks (the swizzle depends on ks); buffer / 32-row block offsets are ds_read immediates
    const char* wb[4];
    const char* xb[4];
    auto late_init = [&]() {
        PIPE_SB();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int rw = wn * (32 * NW) + frow, rx = wm * (32 * MJ) + frow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks * 2 + fq;
            if constexpr (R32) {          // 64-byte rows, two 16-channel sub-steps per K-tile (ks 2, 3 unused)
                wb[ks] = lds + PW_OFF + rw * 64 + (((q & 3) ^ ((rw >> 2) & 3)) << 4);
                xb[ks] = lds + PXO + rx * 64 + (((q & 3) ^ ((rx >> 2) & 3)) << 4);
            } else {
                wb[ks] = lds + PW_OFF + rw * 128 + swz(rw, q) * 16;
                xb[ks] = lds + PXO + rx * 128 + swz(rx, q) * 16;
            }
        }
        PIPE_SB();
    };

    if constexpr (MODE == 2) {
        // ================= role-split stream schedule (round 4) =================
        // What the barrier-level traces of the stream schedule say (profiles/round3_probes/stream_ablations.txt, round4_probes/
        // tile_overhead.txt): a sub-step without copies runs both waves of a SIMD at the full matrix rate (8 MFMAs in ~290 cycles for
        // the leading wave, ~600 for the pair = the pipe's 512), a sub-step in which BOTH waves issue four global->LDS copies takes
        // ~1050 cycles for the same 512 cycles of MFMA work: a wave blocked in vector-memory issue (the CU's copy path takes ~30
        // cycles per 1 KiB piece with four SIMDs feeding it) cannot issue its MFMAs, and its partner is blocked at the same moment.
        // Here the two waves of a SIMD never issue copies at the same time:
        //   waves 0-3 ("X role") issue ALL activation pieces of the tile -- their own rows and their partner's -- in the FIRST half of
        //             a K-tile interval (sub-steps ks3 of the previous tile and ks0),
        //   waves 4-7 ("W role") issue ALL weight pieces in the SECOND half (ks1, ks2),
        // so whichever wave is held up in the copy path, the other one of that SIMD has a full half-tile of MFMAs to feed the pipe
        // with.  The weight pieces then land one whole K-tile later than they do in the stream schedule, which takes a THIRD weight
        // stage: LDS = [W0][W1][W2][X0][X1] = 5 x 32 KiB, all of the CU's 160 KiB (one workgroup per CU anyway).
        // Per K-tile t (X stage t & 1, W stage t % 3), same fragment pipeline as the stream schedule:
        //   ks0: MFMA fa | read F(t,1) | X role: X(t+1) second half (rows 128..255: own + partner pieces) -> X stage 1 - (t&1)
        //   ks1: MFMA fb | read F(t,2) | W role: W(t+2) first half  -> W stage (t+2) % 3   (stage of W(t-1): free since SYNC(t-1))
        //   ks2: MFMA fa | read F(t,3) | W role: W(t+2) second half
        //   SYNC(t): X role vmcnt(0) [X(t+1) landed]; W role vmcnt(8) [everything but the W(t+2) pieces just issued, i.e. W(t+1),
        //            has landed]; lgkmcnt(0); s_barrier
        //   ks3: MFMA fb | read F(t+1,0) | X role: X(t+2) first half -> X stage t & 1 (free behind SYNC(t))
        // RAW: X(t+1) complete at SYNC(t) by the X waves' vmcnt(0) + barrier; W(t+1) was issued during tile t-1 and is complete at
        // SYNC(t) by the W waves' counted wait + barrier; both are first read in ks3(t).  WAR: X stage t&1 is rewritten from ks3(t),
        // behind SYNC(t) (last read F(t,3) in ks2(t), complete at the lgkmcnt(0) of SYNC(t)); W stage of W(t-1) from ks1(t), behind
        // SYNC(t-1).
        const bool xrole = wave < 4;
        const int pw = wave ^ 4;                                         // the SIMD partner (dispatch order 0->2->1->3 twice)
        const int srow_p = pw * 8 + lrow;
        const int b_off0_p = (n0 + srow_p) * a.Cin + swz(srow_p, lchunk) * 8;
        bf16x8 fa[6], fb[6];
        using k0_t = std::integral_constant<int, 0>;
        using k1_t = std::integral_constant<int, 1>;
        using k2_t = std::integral_constant<int, 2>;
        using k3_t = std::integral_constant<int, 3>;
        using k4_t = std::integral_constant<int, 4>;
        using k6_t = std::integral_constant<int, 6>;
        auto rdf = [&](bf16x8 (&f)[6], int woff, auto xbufc, auto ksc, auto firstc, auto lastc) {   // fragments [first, last) of F(., KS)
            constexpr int XB = decltype(xbufc)::value, KS = decltype(ksc)::value, F0 = decltype(firstc)::value, F1 = decltype(lastc)::value;
#pragma unroll
            for (int I = F0; I < F1; ++I) {
                if (I < 2) f[I] = *reinterpret_cast<const bf16x8*>(wb[KS] + woff + I * 4096);
                else f[I] = *reinterpret_cast<const bf16x8*>(xb[KS] + XB * PBX + (I - 2) * 4096);
            }
        };
        auto mmj = [&](bf16x8 (&f)[6], auto jc) {
            constexpr int J = decltype(jc)::value;
            acc[0][J] = LOFT_MFMA_32x32x16(f[0], f[2 + J], acc[0][J]);
            acc[1][J] = LOFT_MFMA_32x32x16(f[1], f[2 + J], acc[1][J]);
        };
        // activation pieces of tile rows i*64 + [wave*8, +8) and i*64 + [partner*8, +8): this wave's row i AND its partner's
        auto issue_xi = [&](auto ic, auto xbufc) {
            constexpr int i = decltype(ic)::value, XB = decltype(xbufc)::value;
            const long aoff = st_aoff + st_c;
            const bf16_t* p0 = ((a_mask[i] >> st_t) & 1u) ? a_ptr[i] + aoff : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p0, (lds_ptr_t)(lds + PXO + XB * PBX + (i * 64 + wave * 8) * 128), 16, 0, 0);
            const bf16_t* p1 = ((a_mask[NI + i] >> st_t) & 1u) ? a_ptr[NI + i] + aoff : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p1, (lds_ptr_t)(lds + PXO + XB * PBX + (i * 64 + pw * 8) * 128), 16, 0, 0);
        };
        // weight pieces of rows (couts) i*64 + [wave*8, +8) and the partner's, into the stage at byte offset woff
        auto issue_wi = [&](auto ic, int woff) {
            constexpr int i = decltype(ic)::value;
            const bf16_t* wt = st_w + sw_c;
            __builtin_amdgcn_global_load_lds((gptr_t)(wt + b_off0 + (long)i * b_step),
                                             (lds_ptr_t)(lds + PW_OFF + woff + (i * 64 + wave * 8) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wt + b_off0_p + (long)i * b_step),
                                             (lds_ptr_t)(lds + PW_OFF + woff + (i * 64 + pw * 8) * 128), 16, 0, 0);
        };
        // ---- prologue (every wave its own pieces, as in the stream schedule): W(0), X(0), W(1); X role: first half of X(1)
        issue_w(c0_t{}, c0_t{}); issue_w(c1_t{}, c0_t{});
        issue_x(c0_t{}, c0_t{}, std::false_type{}); issue_x(c1_t{}, c0_t{}, std::false_type{});
        if (nk > 1) {
            advance_w(); advance_x();
            issue_w(c0_t{}, c1_t{}); issue_w(c1_t{}, c1_t{});
            advance_w();
            if (xrole) {
                issue_xi(k0_t{}, c1_t{}); issue_xi(k1_t{}, c1_t{});
                late_init();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // in flight: W(1) (4 pieces) + the first half of X(1) (4)
            } else {
                late_init();
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // in flight: W(1)
            }
        } else {
            late_init();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PIPE_BARRIER();
        in_loop = true;
        int w_cur = 0, w_nxt = PBW, w_fill = 2 * PBW;                    // byte offsets of the stages of W(t), W(t+1), W(t+2)
        rdf(fa, w_cur, c0_t{}, k0_t{}, k0_t{}, k6_t{});
        // one sub-step: 8 MFMAs on `cur`, the 6 reads of `nxt`, and `copies` (four global->LDS pieces or nothing) pinned between
        // the MFMA pairs
        auto sub = [&](bf16x8 (&cur)[6], bf16x8 (&nxt)[6], int woff, auto xbufc, auto ksc, bool do_read, auto&& copies0, auto&& copies1) {
            PIPE_SB();
            mmj(cur, k0_t{});
            PIPE_SB();
            if (do_read) rdf(nxt, woff, xbufc, ksc, k0_t{}, k2_t{});
            PIPE_SB();
            mmj(cur, k1_t{});
            if (do_read) rdf(nxt, woff, xbufc, ksc, k2_t{}, k4_t{});
            copies0();
            PIPE_SB();
            mmj(cur, k2_t{});
            if (do_read) rdf(nxt, woff, xbufc, ksc, k4_t{}, k6_t{});
            PIPE_SB();
            mmj(cur, k3_t{});
            copies1();
            PIPE_SB();
        };
        auto nop = [] {};
        // the two roles run the same fragment / MFMA stream and differ only in where their copies sit: two loop bodies, selected
        // once per workgroup half (wave-uniform branch; both execute exactly one barrier per K-tile)
        auto rtile = [&](auto rolec, auto xbufc, auto has1, auto has2) {
            constexpr bool XR = decltype(rolec)::value != 0;
            constexpr int XB = decltype(xbufc)::value;
            using xo_t = std::integral_constant<int, 1 - XB>;
            // ks0: X role issues the second half of X(t+1)
            if constexpr (XR)
                sub(fa, fb, w_cur, xbufc, k1_t{}, true, [&] { if (has1) issue_xi(k2_t{}, xo_t{}); },
                    [&] { if (has1) { issue_xi(k3_t{}, xo_t{}); advance_x(); } });
            else
                sub(fa, fb, w_cur, xbufc, k1_t{}, true, nop, nop);
            // ks1, ks2: W role issues W(t+2)
            if constexpr (XR) {
                sub(fb, fa, w_cur, xbufc, k2_t{}, true, nop, nop);
                sub(fa, fb, w_cur, xbufc, k3_t{}, true, nop, nop);
            } else {
                sub(fb, fa, w_cur, xbufc, k2_t{}, true, [&] { if (has2) issue_wi(k0_t{}, w_fill); }, [&] { if (has2) issue_wi(k1_t{}, w_fill); });
                sub(fa, fb, w_cur, xbufc, k3_t{}, true, [&] { if (has2) issue_wi(k2_t{}, w_fill); },
                    [&] { if (has2) { issue_wi(k3_t{}, w_fill); advance_w(); } });
            }
            if (has1) {
                if constexpr (XR) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else {
                    if (has2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                PIPE_BARRIER();
            }
            // ks3: read F(t+1,0); X role issues the first half of X(t+2) into the stage tile t just released
            if constexpr (XR)
                sub(fb, fa, w_nxt, xo_t{}, k0_t{}, has1, [&] { if (has2) issue_xi(k0_t{}, xbufc); }, [&] { if (has2) issue_xi(k1_t{}, xbufc); });
            else
                sub(fb, fa, w_nxt, xo_t{}, k0_t{}, has1, nop, nop);
            const int w_old = w_cur;
            w_cur = w_nxt; w_nxt = w_fill; w_fill = w_old;
        };
        using xr_t = std::integral_constant<int, 1>;
        using wr_t = std::integral_constant<int, 0>;
        if (xrole) {
            int t = 0;
            for (; t + 3 < nk; t += 2) {
                rtile(xr_t{}, c0_t{}, std::true_type{}, std::true_type{});
                rtile(xr_t{}, c1_t{}, std::true_type{}, std::true_type{});
            }
            for (; t < nk; t += 2) {
                rtile(xr_t{}, c0_t{}, t + 1 < nk, t + 2 < nk);
                if (t + 1 < nk) rtile(xr_t{}, c1_t{}, t + 2 < nk, t + 3 < nk);
            }
        } else {
            int t = 0;
            for (; t + 3 < nk; t += 2) {
                rtile(wr_t{}, c0_t{}, std::true_type{}, std::true_type{});
                rtile(wr_t{}, c1_t{}, std::true_type{}, std::true_type{});
            }
            for (; t < nk; t += 2) {
                rtile(wr_t{}, c0_t{}, t + 1 < nk, t + 2 < nk);
                if (t + 1 < nk) rtile(wr_t{}, c1_t{}, t + 2 < nk, t + 3 < nk);
            }
        }
        {
            const bool dense = !a.pixmajor && a.os == 1 && a.OHf == a.OH && a.OWf == a.OW;
            const TileRows otr = pipe_tile_rows<BM>(a, m0, dense, a.OHf, a.OWf, a.Cout, a.os, a.oo_y, a.oo_x);
            if (a.residual) pipe_epilogue_staged<true, MJ, NW>(a, otr, acc, lds, g, m0, n0, wave, lane, ohw, kstamp, a.oo_y, a.oo_x);
            else pipe_epilogue_staged<false, MJ, NW>(a, otr, acc, lds, g, m0, n0, wave, lane, ohw, kstamp, a.oo_y, a.oo_x);
        }
        return;
    }
    if constexpr (MODE == 1) {
        // Static priority for the younger half (round 3): waves 4-7 are dispatched second and lose every VALU / LDS arbitration
        // against their SIMD partner by age -- in the barrier-level trace they reach the K-tile's barrier ~580 cycles after waves
        // 0-3.  One s_setprio for the whole loop (no per-segment flips; MI355X_MICROARCH.md "Two waves per SIMD", item 4):
        // +0.6 .. +2.4 % on every shape of tools/pipe_trace.py, same box (profiles/round3_probes/stream_ablations.txt).
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        // ================= stream schedule =================
        // Per K-tile t (buffer B = t&1), 16-channel sub-steps ks = 0..3; fragment set F(t,ks) = 2 weight + 4 activation
        // ds_read_b128.  Two register sets alternate (ks even -> fa, odd -> fb); the reads of a sub-step are issued ONE SUB-STEP
        // AHEAD of its MFMAs, also across the tile boundary, so the matrix pipe never waits for LDS latency:
        //   ks0: read F(t,1) -> fb | issue X(t+1) -> buffer 1-B (4 copies) | MFMA fa
        //   ks1: read F(t,2) -> fa |                                       | MFMA fb
        //   ks2: read F(t,3) -> fb |                                       | MFMA fa
        //   SYNC: s_waitcnt vmcnt(0) (W(t+1), X(t+1) landed: issued >= 2.5 sub-steps ago), lgkmcnt(0) (this wave's reads of
        //         buffer B are complete), s_barrier                            -- the ONLY barrier of the K-tile
        //   ks3: read F(t+1,0) -> fa (buffer 1-B) | issue W(t+2) -> buffer B (4 copies) | MFMA fb
        // RAW: every piece of K-tile t+1 is covered by the issuing thread's vmcnt(0) and the barrier of SYNC(t) before its first
        // read in ks3(t).  WAR: buffer B is rewritten (W(t+2) in ks3(t), X(t+2) in ks0(t+1)) only behind SYNC(t), by which every
        // wave has completed its last reads of buffer B (F(t,3), issued in ks2(t)).
        bf16x8 fa[6], fb[6];
        auto rd1 = [&](bf16x8 (&f)[6], auto bufc, auto ksc, auto idxc) {      // fragment idx of set F(., KS): 0,1 = W c0,c1; 2..5 = X
            constexpr int B = decltype(bufc)::value, KS = decltype(ksc)::value, I = decltype(idxc)::value;
            if constexpr (I >= 2 + MJ || (I == 1 && NW == 1)) { }              // (smaller tiles: fewer activation / weight fragments)
            else if constexpr (NOREADS) { asm volatile("" : "=v"(f[I])); }
            else if constexpr (I < 2) f[I] = *reinterpret_cast<const bf16x8*>(wb[KS] + B * PBW + I * (R32 ? 2048 : 4096));
            else f[I] = *reinterpret_cast<const bf16x8*>(xb[KS] + B * PBX + (I - 2) * (R32 ? 2048 : 4096));
        };
        auto mm2 = [&](bf16x8 (&f)[6], auto jc) {                               // the two MFMAs of pixel block j
            constexpr int J = decltype(jc)::value;
            if constexpr (J >= MJ) { }
            else if constexpr (NOMFMA) { asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2 + J])); }
            else {
                acc[0][J] = LOFT_MFMA_32x32x16(f[0], f[2 + J], acc[0][J]);
                if constexpr (NW == 2) acc[1][J] = LOFT_MFMA_32x32x16(f[1], f[2 + J], acc[1][J]);
            }
        };
        using k0_t = std::integral_constant<int, 0>;
        using k1_t = std::integral_constant<int, 1>;
        using k2_t = std::integral_constant<int, 2>;
        using k3_t = std::integral_constant<int, 3>;
        using i4_t = std::integral_constant<int, 4>;
        using i5_t = std::integral_constant<int, 5>;
        // one sub-step: 8 MFMAs on `cur` with the 6 reads of `nxt` (and up to four copies) pinned between MFMA pairs -- every
        // other instruction issues in the shadow of an MFMA of this wave instead of in front of the whole cluster
        unsigned long long stp[24];
        int stn = 0, cur_t = 0;
#pragma unroll
        for (int i = 0; i < 24; ++i) stp[i] = 0ull;
        // TRACE: s_memtime into SGPRs with NO wait (a compiler-visible use would force lgkmcnt(0) and distort the time line);
        // the values are stored after the traced tiles behind one explicit wait
#define STREAM_STAMP(slot)                                                                                     \
        do {                                                                                                   \
            if constexpr (TRACE) {                                                                             \
                if (cur_t == 8 || cur_t == 9) asm volatile("s_memtime %0" : "=s"(stp[(slot)]));              \
            }                                                                                                  \
        } while (0)
        auto substep = [&](bf16x8 (&cur)[6], bf16x8 (&nxt)[6], auto bufc, auto ksc, bool do_read, auto&& copy0, auto&& copy1, auto slotc) {
            constexpr int SL = decltype(slotc)::value;
            STREAM_STAMP(SL);
            PIPE_SB();
            mm2(cur, k0_t{});
            PIPE_SB();
            STREAM_STAMP(SL + 1);
            if (do_read) { rd1(nxt, bufc, ksc, k0_t{}); rd1(nxt, bufc, ksc, k1_t{}); }
            PIPE_SB();
            mm2(cur, k1_t{});
            if (do_read) { rd1(nxt, bufc, ksc, k2_t{}); rd1(nxt, bufc, ksc, k3_t{}); }
            copy0();
            PIPE_SB();
            mm2(cur, k2_t{});
            if (do_read) { rd1(nxt, bufc, ksc, i4_t{}); rd1(nxt, bufc, ksc, i5_t{}); }
            PIPE_SB();
            mm2(cur, k3_t{});
            copy1();
            PIPE_SB();
        };
        auto nop = [] {};
        unsigned long long kst_setup = 0ull;
        if constexpr (TRACE) kst_setup = __builtin_amdgcn_s_memtime();
        if constexpr (R32) {
            // ---- ring32 (round 5): 32-channel K-tiles on a FOUR-stage ring.  What the barrier-level traces of the two-stage schedule
            // say (profiles/round3_probes/stream_ablations.txt): its K-tile takes ~3300 cycles for 2048 of MFMA issue because all 64
            // pieces of a K-tile are requested in the two sub-steps behind the barrier (a stage is free only from there on and
            // must have landed one K-tile later), ~1000 cycles each against ~300 for a sub-step without copies.  Half-size K-tiles
            // fit four stages into the same 128 KiB: a tile's pieces are requested THREE tiles (~3000 cycles) ahead, two per wave
            // and sub-step in EVERY sub-step, and retired with a counted vmcnt(6) that never waits for a piece younger than two
            // tiles.  Per K-tile t (stage S = t & 3; 16 MFMAs per wave, two 16-channel sub-steps):
            //   A: MFMA fa = F(t,0) | read F(t,1) -> fb            | request X(t+3) -> stage (S+3) & 3  (free since SYNC(t-1))
            //   SYNC(t): vmcnt(6) [tile t+1 landed; t+2 and X(t+3) in flight], lgkmcnt(0) [own reads of stage S done], s_barrier
            //   B: MFMA fb = F(t,1) | read F(t+1,0) -> fa (stage S+1) | request W(t+3) -> stage (S+3) & 3, advance the sequence
            // The K order is the two-stage schedule's (chunk of 64 channels, tap, half): results are bit-identical to it.
            using z_t = std::integral_constant<int, 0>;
            using st1_t = std::integral_constant<int, 1>;
            using st2_t = std::integral_constant<int, 2>;
            using st3_t = std::integral_constant<int, 3>;
            int sh = 0;                                               // which 32-channel half of the 64-channel chunk is being staged
            auto issue_x32 = [&](auto ic, auto stc) {
                constexpr int i = decltype(ic)::value, ST = decltype(stc)::value;
                const bf16_t* p = ((a_mask[i] >> st_t) & 1u) ? a_ptr[i] + (st_aoff + st_c + sh * 32) : a.zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(lds + PXO + ST * PBX + (wave * 32 + i * 16) * 64), 16, 0, 0);
            };
            auto issue_w32 = [&](auto ic, auto stc) {
                constexpr int i = decltype(ic)::value, ST = decltype(stc)::value;
                const bf16_t* wt = st_w + (sw_c + sh * 32) + b_off0 + (long)i * b_step;
                __builtin_amdgcn_global_load_lds((gptr_t)wt, (lds_ptr_t)(lds + PW_OFF + ST * PBW + (wave * 32 + i * 16) * 64), 16, 0, 0);
            };
            auto advance32 = [&]() {
                sh ^= 1;
                if (sh == 0) { advance_x(); advance_w(); }
            };
            const int nk32 = 2 * nk;
            issue_w32(k0_t{}, z_t{}); issue_w32(k1_t{}, z_t{}); issue_x32(k0_t{}, z_t{}); issue_x32(k1_t{}, z_t{});
            advance32();
            if (nk32 > 1) { issue_w32(k0_t{}, st1_t{}); issue_w32(k1_t{}, st1_t{}); issue_x32(k0_t{}, st1_t{}); issue_x32(k1_t{}, st1_t{}); advance32(); }
            if (nk32 > 2) { issue_w32(k0_t{}, st2_t{}); issue_w32(k1_t{}, st2_t{}); issue_x32(k0_t{}, st2_t{}); issue_x32(k1_t{}, st2_t{}); advance32(); }
            late_init();
            if (nk32 > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (nk32 > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PIPE_BARRIER();
            in_loop = true;
            rd1(fa, z_t{}, k0_t{}, k0_t{}); rd1(fa, z_t{}, k0_t{}, k1_t{}); rd1(fa, z_t{}, k0_t{}, k2_t{});
            rd1(fa, z_t{}, k0_t{}, k3_t{}); rd1(fa, z_t{}, k0_t{}, i4_t{}); rd1(fa, z_t{}, k0_t{}, i5_t{});
            auto tile32 = [&](auto stc, auto has1, auto has3) {
                constexpr int S = decltype(stc)::value;
                using nxt_t = std::integral_constant<int, (S + 1) & 3>;
                using tgt_t = std::integral_constant<int, (S + 3) & 3>;
                substep(fa, fb, stc, k1_t{}, true, [&] { if (has3) issue_x32(k0_t{}, tgt_t{}); }, [&] { if (has3) issue_x32(k1_t{}, tgt_t{}); }, z_t{});
                if (has1) {
                    if (has3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    PIPE_BARRIER();
                }
                substep(fb, fa, nxt_t{}, k0_t{}, has1, [&] { if (has3) issue_w32(k0_t{}, tgt_t{}); },
                        [&] { if (has3) { issue_w32(k1_t{}, tgt_t{}); advance32(); } }, z_t{});
            };
            int t = 0;
            for (; t + 6 < nk32; t += 4) {             // steady state: all four tiles of the round have three successors
                tile32(z_t{}, std::true_type{}, std::true_type{});
                tile32(st1_t{}, std::true_type{}, std::true_type{});
                tile32(st2_t{}, std::true_type{}, std::true_type{});
                tile32(st3_t{}, std::true_type{}, std::true_type{});
            }
            for (; t < nk32; t += 4) {
                tile32(z_t{}, t + 1 < nk32, t + 3 < nk32);
                if (t + 1 < nk32) tile32(st1_t{}, t + 2 < nk32, t + 4 < nk32);
                if (t + 2 < nk32) tile32(st2_t{}, t + 3 < nk32, t + 5 < nk32);
                if (t + 3 < nk32) tile32(st3_t{}, t + 4 < nk32, t + 6 < nk32);
            }
        } else if constexpr (NS == 3) {
            // ---- three-stage ring: per K-tile t (buffer B = t % 3)
            //   ks0: MFMA fa | read F(t,1) | request X(t+2) -> buffer (B+2) % 3      (free since SYNC(t-1): last read by tile t-1)
            //   ks1: MFMA fb | read F(t,2) | request W(t+2) -> buffer (B+2) % 3
            //   ks2: MFMA fa | read F(t,3)
            //   SYNC: s_waitcnt vmcnt(copies of ONE tile) -> tile t+1 (requested during tile t-1) has landed, tile t+2 stays in
            //         flight; lgkmcnt(0); s_barrier
            //   ks3: MFMA fb | read F(t+1,0) from buffer (B+1) % 3
            using c2_t = std::integral_constant<int, 2>;
            using z_t = std::integral_constant<int, 0>;
            constexpr int NXW = NI + 2 * NW;                         // global->LDS copy instructions of one K-tile, per wave
            issue_w(c0_t{}, c0_t{}); issue_w(c1_t{}, c0_t{});
            issue_x(c0_t{}, c0_t{}, std::false_type{}); issue_x(c1_t{}, c0_t{}, std::false_type{});
            if (nk > 1) {
                advance_w(); advance_x();
                issue_w(c0_t{}, c1_t{}); issue_w(c1_t{}, c1_t{});
                issue_x(c0_t{}, c1_t{}, std::false_type{}); issue_x(c1_t{}, c1_t{}, std::false_type{});
                advance_w(); advance_x();
                late_init();
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NXW) : "memory");
            } else {
                late_init();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PIPE_BARRIER();
            in_loop = true;
            rd1(fa, c0_t{}, k0_t{}, k0_t{}); rd1(fa, c0_t{}, k0_t{}, k1_t{}); rd1(fa, c0_t{}, k0_t{}, k2_t{});
            rd1(fa, c0_t{}, k0_t{}, k3_t{}); rd1(fa, c0_t{}, k0_t{}, i4_t{}); rd1(fa, c0_t{}, k0_t{}, i5_t{});
            auto rtile = [&](auto bufc, auto has1, auto has2) {
                constexpr int B = decltype(bufc)::value;
                using next_t = std::integral_constant<int, (B + 1) % 3>;
                using tgt_t = std::integral_constant<int, (B + 2) % 3>;
                substep(fa, fb, bufc, k1_t{}, true,
                        [&] { if (has2) issue_x(c0_t{}, tgt_t{}, std::false_type{}); },
                        [&] { if (has2) { issue_x(c1_t{}, tgt_t{}, std::false_type{}); advance_x(); } }, z_t{});
                substep(fb, fa, bufc, k2_t{}, true,
                        [&] { if (has2) issue_w(c0_t{}, tgt_t{}); },
                        [&] { if (has2) { issue_w(c1_t{}, tgt_t{}); advance_w(); } }, z_t{});
                substep(fa, fb, bufc, k3_t{}, true, nop, nop, z_t{});
                if (has1) {
                    if (has2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NXW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    PIPE_BARRIER();
                }
                substep(fb, fa, next_t{}, k0_t{}, has1, nop, nop, z_t{});
            };
            int t = 0;
            for (; t + 4 < nk; t += 3) {               // steady state: all three tiles of the round have two successors
                rtile(c0_t{}, std::true_type{}, std::true_type{});
                rtile(c1_t{}, std::true_type{}, std::true_type{});
                rtile(c2_t{}, std::true_type{}, std::true_type{});
            }
            for (; t < nk; t += 3) {
                rtile(c0_t{}, t + 1 < nk, t + 2 < nk);
                if (t + 1 < nk) rtile(c1_t{}, t + 2 < nk, t + 3 < nk);
                if (t + 2 < nk) rtile(c2_t{}, t + 3 < nk, t + 4 < nk);
            }
        } else {
        // prologue: W(0), X(0), W(1)
        // (XF, round 6 -- what the barrier-level traces of round 3 show is that the ACTIVATION copies are the ones waited for: X(t+1)
        //  is requested in ks0(t), 2.5 sub-steps before SYNC(t), and a tile whose activation copies are the only ones in flight
        //  waits ~760 cycles for them there -- gathered rows of 256 different RoIs / pixels, each its own line, against weight rows
        //  that every CU of the XCD fetches at the same moment and finds in L2.  XF swaps the two operands' places in the
        //  schedule: X(t+2) is requested in ks3(t), right behind SYNC(t), and has a whole K-tile to land; W(t+1) is requested in
        //  ks0(t).  Prologue: W(0), X(0), X(1).  RAW / WAR: the mirror image of the argument above -- W(t+1) -> buffer 1-B in ks0(t):
        //  that buffer's last reads (F(t-1,3)) completed at SYNC(t-1); X(t+2) -> buffer B in ks3(t): behind SYNC(t).  Same K order:
        //  bit-identical results.)
        issue_w(c0_t{}, c0_t{}); issue_w(c1_t{}, c0_t{});
        issue_x(c0_t{}, c0_t{}, std::false_type{}); issue_x(c1_t{}, c0_t{}, std::false_type{});
        if (nk > 1 && XF) {
            advance_x();
            issue_x(c0_t{}, c1_t{}, std::false_type{}); issue_x(c1_t{}, c1_t{}, std::false_type{});
            advance_x(); advance_w();
            late_init();
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything but X(1)'s copies has landed
        } else if (nk > 1) {
            advance_w(); advance_x();
            issue_w(c0_t{}, c1_t{}); issue_w(c1_t{}, c1_t{});
            advance_w();
            late_init();
            if constexpr (NW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything but W(1)'s copies has landed
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            late_init();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PIPE_BARRIER();
        in_loop = true;
#pragma unroll
        for (int ks = 0; ks < 1; ++ks) {
            rd1(fa, c0_t{}, k0_t{}, k0_t{}); rd1(fa, c0_t{}, k0_t{}, k1_t{}); rd1(fa, c0_t{}, k0_t{}, k2_t{});
            rd1(fa, c0_t{}, k0_t{}, k3_t{}); rd1(fa, c0_t{}, k0_t{}, i4_t{}); rd1(fa, c0_t{}, k0_t{}, i5_t{});
        }
        // has1 / has2 (K-tiles t+1 / t+2 exist): std::true_type in the steady-state loop -- the conditions fold away --, bool in the tail
        auto stile = [&](auto bufc, auto has1, auto has2, auto avc) {
            constexpr int B = decltype(bufc)::value;
            using other_t = std::integral_constant<int, 1 - B>;
            // ---- ks0: MFMA fa, read F(t,1) -> fb, issue X(t+1) -> buffer 1-B
            using s0_t = std::integral_constant<int, 12 * B + 0>;
            using s1_t = std::integral_constant<int, 12 * B + 2>;
            using s2_t = std::integral_constant<int, 12 * B + 4>;
            using s3_t = std::integral_constant<int, 12 * B + 8>;
            const bool copier = !ALTCOPY || (wave >> 2) == B;
            if constexpr (XF)
                substep(fa, fb, bufc, k1_t{}, true, [&] { if (has1) issue_w(c0_t{}, other_t{}); },
                        [&] { if (has1) { issue_w(c1_t{}, other_t{}); advance_w(); } }, s0_t{});
            else
            substep(fa, fb, bufc, k1_t{}, true,
                    [&] { if (has1 && copier) { issue_x(c0_t{}, other_t{}, avc); if constexpr (ALTCOPY) issue_x(c0_t{}, other_t{}, avc); } },
                    [&] { if (has1) { if (copier) { issue_x(c1_t{}, other_t{}, avc); if constexpr (ALTCOPY) issue_x(c1_t{}, other_t{}, avc); } advance_x(); } }, s0_t{});
            // ---- ks1, ks2
            substep(fb, fa, bufc, k2_t{}, true, nop, nop, s1_t{});
            substep(fa, fb, bufc, k3_t{}, true, nop, nop, s2_t{});
            // ---- SYNC
            if (has1) {
                STREAM_STAMP(12 * B + 6);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                STREAM_STAMP(12 * B + 7);
                PIPE_BARRIER();
            }
            // ---- ks3: MFMA fb, read F(t+1,0) -> fa from buffer 1-B, issue W(t+2) -> buffer B
            if constexpr (XF)
                substep(fb, fa, other_t{}, k0_t{}, has1, [&] { if (has2) issue_x(c0_t{}, bufc, avc); },
                        [&] { if (has2) { issue_x(c1_t{}, bufc, avc); advance_x(); } }, s3_t{});
            else
            substep(fb, fa, other_t{}, k0_t{}, has1,
                    [&] { if (has2 && copier) { issue_w(c0_t{}, bufc); if constexpr (ALTCOPY) issue_w(c0_t{}, bufc); } },
                    [&] { if (has2) { if (copier) { issue_w(c1_t{}, bufc); if constexpr (ALTCOPY) issue_w(c1_t{}, bufc); } advance_w(); } }, s3_t{});
            STREAM_STAMP(12 * B + 10);
            ++cur_t;
            if constexpr (TRACE) {
                if (cur_t == 10 && (int)blockIdx.x >= trace_b0 && (int)blockIdx.x < trace_b0 + 2 && blockIdx.y == 0 && blockIdx.z == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    PIPE_SB();
                    unsigned long long* tr = reinterpret_cast<unsigned long long*>(a.trace) + ((long)(blockIdx.x - trace_b0) * 8 + wave) * 64;
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 24; ++i) tr[i] = stp[i];
                        unsigned hwid;
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                        tr[63] = hwid;
                    }
                }
            }
        };
        if constexpr (TRACE) kst1 = __builtin_amdgcn_s_memtime();
        auto run_tiles = [&](auto avc) {
            int t = 0;
            if constexpr (!TRACE) {
                for (; t + 3 < nk; t += 2) {          // steady state: both tiles of the pair have two successors
                    stile(c0_t{}, std::true_type{}, std::true_type{}, avc);
                    stile(c1_t{}, std::true_type{}, std::true_type{}, avc);
                }
            }
            for (; t < nk; t += 2) {
                stile(c0_t{}, t + 1 < nk, t + 2 < nk, avc);
                if (t + 1 < nk) stile(c1_t{}, t + 2 < nk, t + 3 < nk, avc);
            }
        };
        if constexpr (LEAN) {              // (the all-valid form is its own copy of the loop: no branch inside the K-tile)
            if (all_valid) run_tiles(std::true_type{});
            else run_tiles(std::false_type{});
        } else run_tiles(std::false_type{});
        }
        if constexpr (TRACE) kst2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(0);
        if constexpr (VAR & 2)
            conv_epilogue<2, 4, 128, 64>(a, acc, g, m0, n0, wave >> 2, wave & 3, lane & 31, lane >> 5, ohw, nullptr, nullptr, nullptr,
                                         a.pixmajor != 0);
        else {
            const bool dense = !a.pixmajor && a.os == 1 && a.OHf == a.OH && a.OWf == a.OW;       // output pixel index == m
            // par_n (loft_deconv2x2_bf16): the launch's N tiles are the FOUR OUTPUT PARITIES of a 2x2 / stride-2 deconvolution -- N
            // tile p holds the 256 output channels of tap p and stores them at output offset (p >> 1, p & 1) of the stride-2 grid;
            // the four tiles of one pixel tile run back to back on one XCD (nfast), so the input tile comes from HBM once
            const int par = a.par_n ? by : 0, n0o = a.par_n ? 0 : n0;
            const int ooy = a.oo_y + (par >> 1), oox = a.oo_x + (par & 1);
            const TileRows otr = pipe_tile_rows<BM>(a, m0, dense, a.OHf, a.OWf, a.Cout, a.os, ooy, oox);
            if constexpr (PL) {
                if constexpr (MJ == 4 && NW == 2) {
                    if (a.staged_out && otr.lin) pipe_epilogue_f32_staged(a, otr, acc, lds, g, n0, wave, lane);
                    else pipe_epilogue_f32<MJ, NW>(a, otr, acc, g, m0, n0, wave, lane, ohw);
                } else pipe_epilogue_f32<MJ, NW>(a, otr, acc, g, m0, n0, wave, lane, ohw);
            }
            else if (a.residual) pipe_epilogue_staged<true, MJ, NW>(a, otr, acc, lds, g, m0, n0o, wave, lane, ohw, kstamp, ooy, oox);
            else pipe_epilogue_staged<false, MJ, NW>(a, otr, acc, lds, g, m0, n0o, wave, lane, ohw, kstamp, ooy, oox);
        }
        if constexpr (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long kst3 = __builtin_amdgcn_s_memtime();
            if ((int)blockIdx.x >= trace_b0 && (int)blockIdx.x < trace_b0 + 2 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
                unsigned long long* tr = reinterpret_cast<unsigned long long*>(a.trace) + ((long)(blockIdx.x - trace_b0) * 8 + wave) * 64;
                tr[32] = kst0; tr[33] = kst1; tr[34] = kst2; tr[35] = kst3; tr[36] = kst_setup;
            }
        }
        return;
    }
    // ---- prologue: all of K-tile 0 and W0 of K-tile 1
    issue_w(c0_t{}, c0_t{});
    issue_w(c1_t{}, c0_t{});
    issue_x(c0_t{}, c0_t{}, std::false_type{});
    issue_x(c1_t{}, c0_t{}, std::false_type{});
    if (nk > 1) {
        advance();
        issue_w(c0_t{}, c1_t{});
        late_init();
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        late_init();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PIPE_BARRIER();
    if (wm == 1) PIPE_BARRIER();          // the stagger: group 1 runs one slot behind group 0

    bf16x8 wf0[4];                       // W c0 of the CURRENT K-tile: read one phase early (previous tile's L3)
    unsigned long long* trace = nullptr;
    int trace_n = 0;
    if constexpr (TRACE) {
        trace = reinterpret_cast<unsigned long long*>(a.trace) + ((long)(blockIdx.x == 0 ? 0 : 1) * 8 + wave) * 64;
        if (blockIdx.x < 2 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            trace[63] = hwid;
        }
    }
    auto stamp = [&](int t) {
        if constexpr (TRACE) {
            if (t >= 8 && t < 11 && blockIdx.x < 2 && blockIdx.y == 0 && blockIdx.z == 0 && trace_n < 63) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if (lane == 0) trace[trace_n] = now;
                ++trace_n;
            }
        }
    };
#define PIPE_PRIO(p) do { if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(p); } while (0)
    if constexpr (!OLDORDER) {
        // W c0 of K-tile 0 (retired by the prologue wait + barrier above)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf0[ks] = *reinterpret_cast<const bf16x8*>(wb[ks]);
    }

    in_loop = true;
    auto rd = [&](const char* p) {
        if constexpr (NOREADS) { bf16x8 v; asm volatile("" : "=v"(v)); return v; }
        else return *reinterpret_cast<const bf16x8*>(p);
    };
    auto mm = [&](bf16x8 w_, bf16x8 x_, f32x16 c_) {
        if constexpr (NOMFMA) { asm volatile("" :: "v"(w_), "v"(x_)); return c_; }
        else return LOFT_MFMA_32x32x16(w_, x_, c_);
    };
    auto tile = [&](auto bufc, int t, bool has1, bool has2) {
        constexpr int B = decltype(bufc)::value;
        using other_t = std::integral_constant<int, 1 - B>;
        bf16x8 xf[2][4], wf1[4], wfn[4];
        // ---------------- L0: X p0 (+ W c0 in the old order); issue W1(t+1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (OLDORDER) wf0[ks] = rd(wb[ks] + B * PBUF);
            xf[0][ks] = rd(xb[ks] + B * PBUF);
            xf[1][ks] = rd(xb[ks] + B * PBUF + 4096);
        }
        if (has1) issue_w(c1_t{}, other_t{});
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- M0
        if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIPE_SB(); stamp(t); }
        PIPE_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = mm(wf0[ks], xf[0][ks], acc[0][0]);
            acc[0][1] = mm(wf0[ks], xf[1][ks], acc[0][1]);
        }
        PIPE_PRIO(0);
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- L1: W c1; issue X0(t+1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf1[ks] = rd(wb[ks] + B * PBUF + 4096);
        if (has1) issue_x(c0_t{}, other_t{}, std::false_type{});
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- M1
        if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIPE_SB(); stamp(t); }
        PIPE_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[1][0] = mm(wf1[ks], xf[0][ks], acc[1][0]);
            acc[1][1] = mm(wf1[ks], xf[1][ks], acc[1][1]);
        }
        PIPE_PRIO(0);
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- L2: X p1 (same registers); issue X1(t+1); retire W1(t+1) (new order: it is read in L3)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xf[0][ks] = rd(xb[ks] + B * PBUF + 8192);
            xf[1][ks] = rd(xb[ks] + B * PBUF + 12288);
        }
        if (has1) issue_x(c1_t{}, other_t{}, std::false_type{});
        if constexpr (!OLDORDER) {
            if (has1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // in flight: X0(t+1), X1(t+1)
        }
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- M2
        if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIPE_SB(); stamp(t); }
        PIPE_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[1][2] = mm(wf1[ks], xf[0][ks], acc[1][2]);
            acc[1][3] = mm(wf1[ks], xf[1][ks], acc[1][3]);
        }
        PIPE_PRIO(0);
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- L3: (new order) W c0 of K-tile t+1; advance the staging state, issue W0(t+2); retire X0(t+1)
        if constexpr (!OLDORDER) {
            if (has1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wfn[ks] = rd(wb[ks] + (1 - B) * PBUF);
            }
        }
        if (has2) {
            advance();
            issue_w(c0_t{}, bufc);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
        // ---------------- M3
        if constexpr (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIPE_SB(); stamp(t); }
        PIPE_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][2] = mm(wf0[ks], xf[0][ks], acc[0][2]);
            acc[0][3] = mm(wf0[ks], xf[1][ks], acc[0][3]);
        }
        PIPE_PRIO(0);
        if (has2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!OLDORDER) {
            if (has1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wf0[ks] = wfn[ks];
            }
        }
        stamp(t);
        PIPE_BARRIER();
        stamp(t);
    };

    for (int t = 0; t < nk; t += 2) {
        tile(c0_t{}, t, t + 1 < nk, t + 2 < nk);
        if (t + 1 < nk) tile(c1_t{}, t + 1, t + 2 < nk, t + 3 < nk);
    }
    if (wm == 0) PIPE_BARRIER();          // barrier counts of the two groups match again

    {
        const bool dense = !a.pixmajor && a.os == 1 && a.OHf == a.OH && a.OWf == a.OW;
        const TileRows otr = pipe_tile_rows<256>(a, m0, dense, a.OHf, a.OWf, a.Cout, a.os, a.oo_y, a.oo_x);
        if (a.residual) pipe_epilogue_staged<true, 4, 2>(a, otr, acc, lds, g, m0, n0, wave, lane, ohw, kstamp, a.oo_y, a.oo_x);
        else pipe_epilogue_staged<false, 4, 2>(a, otr, acc, lds, g, m0, n0, wave, lane, ohw, kstamp, a.oo_y, a.oo_x);
    }
}

// =====================================================================================
// conv_tap_w4_kernel (round 5, LOFT_CONV_W4): the stream schedule's 256 x 256 x 64 tile with FOUR waves -- ONE per SIMD, each
// owning 128 pixels x 128 couts (16 accumulator blocks = 256 registers, the kernel takes the whole 512-register file).  Why: in
// the eight-wave kernel the two waves of a SIMD share its matrix pipe, and the barrier-level traces show both of them in their
// copy sub-steps (and at the K-tile's barrier) at the same moment -- ~1000 cycles for 512 cycles of MFMA work twice per K-tile.
// A single wave per SIMD has nobody to collide with: its 64 MFMAs per K-tile issue back to back while its 32 fragment reads and
// 16 copies sit in the gaps (<= 2 other instructions per MFMA).  Same LDS layout, staging, swizzle, tap sequence, K order and
// two-stage ring as conv_tap_pipe_kernel<1,0,4,2>: results are bit-identical to it.  EPI 0: direct 16-bit epilogue.
// Reference call sites: as conv_tap_pipe_kernel (FOA / mask / FPN / RPN 3x3 layers, the shared FCs).
// =====================================================================================
template <int EPI>
__global__ __launch_bounds__(256) void conv_tap_w4_kernel(const ConvArgs a) {
    constexpr int BM = 256, BN = 256, NI = 8;
    __shared__ __attribute__((aligned(16))) char lds[PLDS];
    asm volatile("" :: "s"(a.src), "s"(a.wgt), "s"(a.zero_page), "s"(a.out), "s"(a.src_gs), "s"(a.wgt_gs), "s"(a.gxy_mul), "s"(a.gxy_sh),
                 "s"(a.gx_mul), "s"(a.gx_sh), "s"(a.gy_mul), "s"(a.gy_sh), "s"(a.nfast), "s"(a.pixmajor), "s"(a.pointwise), "s"(a.T),
                 "s"(a.B), "s"(a.IH), "s"(a.IW), "s"(a.Cin), "s"(a.Cout), "s"(a.OH), "s"(a.OW), "s"(a.M), "s"(a.ss), "s"(a.pm_S),
                 "s"(a.pm_P), "s"(a.pms_mul), "s"(a.pms_sh));
    asm volatile("" :: "s"(a.pmp_mul), "s"(a.pmp_sh), "s"(a.ohw_mul), "s"(a.ohw_sh), "s"(a.ow_mul), "s"(a.ow_sh),
                 "s"(a.dy_pk), "s"(a.dx_pk), "s"(a.wt_pk), "s"(a.pk_ok));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bz = fastdiv(V, a.gxy_mul, a.gxy_sh), Vg = V - bz * (int)(gridDim.x * gridDim.y);
    int bx, by;
    if (a.nfast) { bx = fastdiv(Vg, a.gy_mul, a.gy_sh); by = Vg - bx * (int)gridDim.y; }
    else { by = fastdiv(Vg, a.gx_mul, a.gx_sh); bx = Vg - by * (int)gridDim.x; }
    const int m0 = bx * BM, n0 = by * BN, g = bz;
    const bf16_t* src = a.src + (long)g * a.src_gs;
    const bf16_t* wgt = a.wgt + (long)g * a.wgt_gs;
    const int lrow = lane >> 3, lchunk = lane & 7;
    const int ohw = a.OH * a.OW;
    const int kchunks = a.Cin / BK;
    // staging: thread -> tile rows i*32 + wave*8 + lrow, i = 0..7, of both operands (rows 32 apart share the swizzle)
    const int srow = wave * 8 + lrow;
    const int schk = swz(srow, lchunk) * 8;
    const int b_off0 = (n0 + srow) * a.Cin + schk;
    const int b_step = 32 * a.Cin;
    int tab_dy = 0, tab_dx = 0, tab_a = 0, tab_w = 0;
    if (a.pointwise) tab_w = a.wt[0] * a.Cout * a.Cin;
    else if (a.pk_ok) {
        const int sh = (lane & 15) * 4;
        tab_dy = (int)((a.dy_pk >> sh) & 15ull) - 8; tab_dx = (int)((a.dx_pk >> sh) & 15ull) - 8;
        tab_a = (tab_dy * a.IW + tab_dx) * a.Cin;
        tab_w = (int)((a.wt_pk >> sh) & 15ull) * a.Cout * a.Cin;
    } else if (lane < a.T) {
        tab_dy = a.dy[lane]; tab_dx = a.dx[lane];
        tab_a = (tab_dy * a.IW + tab_dx) * a.Cin;
        tab_w = a.wt[lane] * a.Cout * a.Cin;
    }
    const bf16_t* a_ptr[NI];
    unsigned a_mask[NI];
    unsigned tmask = 0xffffffffu;
    if (a.pixmajor) {
        int rs1, b00, b10, pos0, pos1;
        pipe_pm_tile(a, m0, rs1, b00, b10, pos0, pos1);
        const int oy0 = fastdiv(pos0, a.ow_mul, a.ow_sh), ox0 = pos0 - oy0 * a.OW;
        const int oy1 = fastdiv(pos1, a.ow_mul, a.ow_sh), ox1 = pos1 - oy1 * a.OW;
        unsigned mk0 = 0u, mk1 = 0u;
        for (int t = 0; t < a.T; ++t) {
            const int dy = __builtin_amdgcn_readlane(tab_dy, t), dx = __builtin_amdgcn_readlane(tab_dx, t);
            const int iy0_ = oy0 * a.ss + dy, ix0_ = ox0 * a.ss + dx, iy1_ = oy1 * a.ss + dy, ix1_ = ox1 * a.ss + dx;
            mk0 |= ((iy0_ >= 0) & (iy0_ < a.IH) & (ix0_ >= 0) & (ix0_ < a.IW)) ? (1u << t) : 0u;
            mk1 |= ((iy1_ >= 0) & (iy1_ < a.IH) & (ix1_ >= 0) & (ix1_ < a.IW)) ? (1u << t) : 0u;
        }
        const int rows = pipe_clamp(a.M - m0, 0, BM);
        const int nv0 = pipe_clamp(a.B - b00, 0, rs1 < rows ? rs1 : rows);
        const int nv1 = pipe_clamp(a.B - b10, 0, rows - rs1 > 0 ? rows - rs1 : 0);
        const long in0 = ((long)(b00 * a.IH + oy0 * a.ss) * a.IW + ox0 * a.ss) * a.Cin;
        const long in1 = ((long)(b10 * a.IH + oy1 * a.ss) * a.IW + ox1 * a.ss) * a.Cin;
        const long istr = (long)a.IH * a.IW * a.Cin;
        tmask = (nv0 > 0 ? mk0 : 0u) | (nv1 > 0 ? mk1 : 0u);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * 32 + srow;
            const bool k = row >= rs1;
            const int idx = k ? row - rs1 : row;
            const bool ok = idx < (k ? nv1 : nv0);
            a_ptr[i] = ok ? src + ((k ? in1 : in0) + (long)idx * istr + schk) : src;
            a_mask[i] = ok ? (k ? mk1 : mk0) : 0u;
        }
    } else {
        int a_iy[NI], a_ix[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = m0 + i * 32 + srow;
            a_ptr[i] = src;
            a_mask[i] = 0u;
            a_iy[i] = -(1 << 20); a_ix[i] = -(1 << 20);
            if (a.pointwise) {
                if (m < a.M) { a_mask[i] = 1u; a_ptr[i] = src + ((long)m * a.Cin + schk); }
                continue;
            }
            if (m < a.M) {
                const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                a_iy[i] = oy * a.ss; a_ix[i] = ox * a.ss;
                a_ptr[i] = src + ((long)(b * a.IH * a.IW + a_iy[i] * a.IW + a_ix[i]) * a.Cin + schk);
            }
        }
        if (!a.pointwise) {
            for (int t = 0; t < a.T; ++t) {
                const int dy = __builtin_amdgcn_readlane(tab_dy, t), dx = __builtin_amdgcn_readlane(tab_dx, t);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                    a_mask[i] |= ((iy >= 0) & (iy < a.IH) & (ix >= 0) & (ix < a.IW)) ? (1u << t) : 0u;
                }
            }
        }
    }
    // the taps this tile needs, compacted into lanes (see conv_tap_pipe_kernel); chunk-major K order
    int vt_a = 0, vt_w = 0, vt_m = 0, nv = 0;
    for (int t = 0; t < a.T; ++t) {
        if ((tmask >> t) & 1u) {
            const int ta = __builtin_amdgcn_readlane(tab_a, t), tw = __builtin_amdgcn_readlane(tab_w, t);
            vt_a = lane == nv ? ta : vt_a;
            vt_w = lane == nv ? tw : vt_w;
            vt_m = lane == nv ? t : vt_m;
            ++nv;
        }
    }
    if (a.pointwise) { vt_w = tab_w; nv = 1; }
    const int nk = nv * kchunks;
    int xj = 0, wj = 0, st_c = 0, sw_c = 0;
    int st_t = __builtin_amdgcn_readlane(vt_m, 0);
    long st_aoff = (long)__builtin_amdgcn_readlane(vt_a, 0);
    const bf16_t* st_w = wgt + (long)__builtin_amdgcn_readlane(vt_w, 0);
    auto seq_step = [&](int& j, int& c) {
        const int jn = j + 1, cn = c + BK;
        const bool wj_ = jn >= nv, wc_ = cn == a.Cin;
        c = wj_ ? (wc_ ? 0 : cn) : c;
        j = wj_ ? 0 : jn;
    };
    auto advance_w = [&]() { seq_step(wj, sw_c); st_w = wgt + (long)__builtin_amdgcn_readlane(vt_w, wj); };
    auto advance_x = [&]() {
        seq_step(xj, st_c);
        st_t = __builtin_amdgcn_readlane(vt_m, xj);
        st_aoff = (long)__builtin_amdgcn_readlane(vt_a, xj);
    };
    // pieces i0, i0 + 1 of the weight / activation tile into buffer B
    auto issue_w = [&](auto i0c, auto bufc) {
        constexpr int I0 = decltype(i0c)::value, B = decltype(bufc)::value;
        const bf16_t* wt = st_w + sw_c + b_off0;
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wt + (long)i * b_step), (lds_ptr_t)(lds + PW_OFF + B * PBUF + (i * 32 + wave * 8) * 128), 16, 0, 0);
    };
    auto issue_x = [&](auto i0c, auto bufc) {
        constexpr int I0 = decltype(i0c)::value, B = decltype(bufc)::value;
        const long aoff = st_aoff + st_c;
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i) {
            const bf16_t* p = ((a_mask[i] >> st_t) & 1u) ? a_ptr[i] + aoff : a.zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(lds + PX_OFF + B * PBUF + (i * 32 + wave * 8) * 128), 16, 0, 0);
        }
    };
    using c0_t = std::integral_constant<int, 0>;
    using c1_t = std::integral_constant<int, 1>;
    using c2_t = std::integral_constant<int, 2>;
    using c3_t = std::integral_constant<int, 3>;
    using c4_t = std::integral_constant<int, 4>;
    using c6_t = std::integral_constant<int, 6>;
    f32x16 acc[4][4];
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fq = lane >> 5;
    const char* wb[4];
    const char* xb[4];
    // prologue: W(0), X(0), W(1)
    issue_w(c0_t{}, c0_t{}); issue_w(c2_t{}, c0_t{}); issue_w(c4_t{}, c0_t{}); issue_w(c6_t{}, c0_t{});
    issue_x(c0_t{}, c0_t{}); issue_x(c2_t{}, c0_t{}); issue_x(c4_t{}, c0_t{}); issue_x(c6_t{}, c0_t{});
    if (nk > 1) {
        advance_w(); advance_x();
        issue_w(c0_t{}, c1_t{}); issue_w(c2_t{}, c1_t{}); issue_w(c4_t{}, c1_t{}); issue_w(c6_t{}, c1_t{});
        advance_w();
    }
    PIPE_SB();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
        const int rw = wn * 128 + frow, rx = wm * 128 + frow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks * 2 + fq;
            wb[ks] = lds + PW_OFF + rw * 128 + swz(rw, q) * 16;
            xb[ks] = lds + PX_OFF + rx * 128 + swz(rx, q) * 16;
        }
    }
    PIPE_SB();
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // everything but W(1)'s eight copies has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PIPE_BARRIER();
    bf16x8 fa[8], fb[8];
    auto rd1 = [&](bf16x8 (&f)[8], auto bufc, auto ksc, auto idxc) {      // fragment idx of F(., KS): 0..3 = W blocks, 4..7 = X blocks
        constexpr int B = decltype(bufc)::value, KS = decltype(ksc)::value, I = decltype(idxc)::value;
        if constexpr (I < 4) f[I] = *reinterpret_cast<const bf16x8*>(wb[KS] + B * PBUF + I * 4096);
        else f[I] = *reinterpret_cast<const bf16x8*>(xb[KS] + B * PBUF + (I - 4) * 4096);
    };
    auto mm4 = [&](bf16x8 (&f)[8], auto jc) {                               // the four MFMAs of pixel block j
        constexpr int J = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][J] = LOFT_MFMA_32x32x16(f[i], f[4 + J], acc[i][J]);
    };
    using i5_t = std::integral_constant<int, 5>;
    using i7_t = std::integral_constant<int, 7>;
    // one sub-step: 16 MFMAs on `cur`, the 8 reads of `nxt` and up to eight copies pinned between the groups of four MFMAs
    auto substep = [&](bf16x8 (&cur)[8], bf16x8 (&nxt)[8], auto bufc, auto ksc, bool do_read, auto&& cp0, auto&& cp1, auto&& cp2, auto&& cp3) {
        PIPE_SB();
        mm4(cur, c0_t{});
        if (do_read) { rd1(nxt, bufc, ksc, c0_t{}); rd1(nxt, bufc, ksc, c1_t{}); }
        cp0();
        PIPE_SB();
        mm4(cur, c1_t{});
        if (do_read) { rd1(nxt, bufc, ksc, c2_t{}); rd1(nxt, bufc, ksc, c3_t{}); }
        cp1();
        PIPE_SB();
        mm4(cur, c2_t{});
        if (do_read) { rd1(nxt, bufc, ksc, c4_t{}); rd1(nxt, bufc, ksc, i5_t{}); }
        cp2();
        PIPE_SB();
        mm4(cur, c3_t{});
        if (do_read) { rd1(nxt, bufc, ksc, c6_t{}); rd1(nxt, bufc, ksc, i7_t{}); }
        cp3();
        PIPE_SB();
    };
    auto nop = [] {};
    rd1(fa, c0_t{}, c0_t{}, c0_t{}); rd1(fa, c0_t{}, c0_t{}, c1_t{}); rd1(fa, c0_t{}, c0_t{}, c2_t{}); rd1(fa, c0_t{}, c0_t{}, c3_t{});
    rd1(fa, c0_t{}, c0_t{}, c4_t{}); rd1(fa, c0_t{}, c0_t{}, i5_t{}); rd1(fa, c0_t{}, c0_t{}, c6_t{}); rd1(fa, c0_t{}, c0_t{}, i7_t{});
    // per K-tile t (buffer B = t & 1): ks0 MFMA fa | read F(t,1) | X(t+1) -> buffer 1-B;  ks1, ks2;  SYNC (vmcnt(0), lgkmcnt(0),
    // barrier);  ks3 MFMA fb | read F(t+1,0) from buffer 1-B | W(t+2) -> buffer B   (RAW / WAR: as the eight-wave stream schedule)
    auto stile = [&](auto bufc, auto has1, auto has2) {
        constexpr int B = decltype(bufc)::value;
        using other_t = std::integral_constant<int, 1 - B>;
        substep(fa, fb, bufc, c1_t{}, true,
                [&] { if (has1) issue_x(c0_t{}, other_t{}); }, [&] { if (has1) issue_x(c2_t{}, other_t{}); },
                [&] { if (has1) issue_x(c4_t{}, other_t{}); }, [&] { if (has1) { issue_x(c6_t{}, other_t{}); advance_x(); } });
        substep(fb, fa, bufc, c2_t{}, true, nop, nop, nop, nop);
        substep(fa, fb, bufc, c3_t{}, true, nop, nop, nop, nop);
        if (has1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            PIPE_BARRIER();
        }
        substep(fb, fa, other_t{}, c0_t{}, has1,
                [&] { if (has2) issue_w(c0_t{}, bufc); }, [&] { if (has2) issue_w(c2_t{}, bufc); },
                [&] { if (has2) issue_w(c4_t{}, bufc); }, [&] { if (has2) { issue_w(c6_t{}, bufc); advance_w(); } });
    };
    {
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            stile(c0_t{}, std::true_type{}, std::true_type{});
            stile(c1_t{}, std::true_type{}, std::true_type{});
        }
        for (; t < nk; t += 2) {
            stile(c0_t{}, t + 1 < nk, t + 2 < nk);
            if (t + 1 < nk) stile(c1_t{}, t + 2 < nk, t + 3 < nk);
        }
    }
    // ---- direct 16-bit epilogue: lane = pixel row wm*128 + j*32 + frow, four consecutive couts per (i, gq)
    {
        const bool dense = !a.pixmajor && a.os == 1 && a.OHf == a.OH && a.OWf == a.OW;
        const TileRows tr = pipe_tile_rows<BM>(a, m0, dense, a.OHf, a.OWf, a.Cout, a.os, a.oo_y, a.oo_x);
        const long out_g = (long)g * a.out_gs;
        const float* bias = a.bias ? a.bias + (long)g * a.bias_gs : nullptr;
        const bf16_t* res = a.residual;
        const bf16_t* msk = a.mask;
        bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
        const bool relu = a.relu != 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wm * 128 + j * 32 + frow;
            long roff;
            bool ok;
            if (tr.lin) roff = pipe_row_off(tr, r, ok);
            else {
                const int m = m0 + r;
                ok = m < a.M;
                roff = 0;
                if (ok) {
                    int b, rem;
                    pipe_row_decode(a, m, ohw, b, rem);
                    ok = b < a.B;
                    const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                    roff = (((long)b * a.OHf + oy * a.os + a.oo_y) * a.OWf + ox * a.os + a.oo_x) * a.Cout;
                }
            }
            if (!ok) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = n0 + wn * 128 + i * 32 + 8 * gq + 4 * fq;
                    const long o = out_g + roff + n;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][gq * 4 + e];
                    if (bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    }
                    if (res) {
                        float rv[4];
                        ld4(res + o, rv);
                        v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (msk) {
                        float mv[4];
                        ld4(msk + o, mv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                    }
                    st4(out + o, v);
                }
        }
    }
}

// Which form of the two-stage 256 x 256 stream schedule the DISPATCHER launches (explicit LOFT_CONV_* requests are unaffected):
// 0 the round-2 schedule, 1 activations-first (LOFT_CONV_XFIRST), 2 lean (LOFT_CONV_LEAN), 3 both (LOFT_CONV_LEANX).  All four
// compute the same sums in the same order (bit-identical); the setter exists so that one process can time a whole training step
// under each (tools/ab_stream_form.sh).  Not an environment switch: the library reads no environment.
static int g_stream_form = LOFT_STREAM_FORM_DEFAULT;
static int g_f32_direct_epilogue = 0;       // (form bit 2: plane launches keep the direct, un-staged fp32 epilogue -- A/B of the round-6 form)
LOFT_EXPORT int loft_conv_stream_form(int form) {
    const int prev = g_stream_form | (g_f32_direct_epilogue << 2);
    if (form >= 0 && form <= 7) { g_stream_form = form & 3; g_f32_direct_epilogue = (form >> 2) & 1; }
    return prev;
}

// host side: launched from loft_conv_tap_bf16_v (conv_mfma.hip).  Requires Cout % 256 == 0, Cin % 64 == 0, T <= 16.
int loft_launch_conv_tap_pipe(const ConvArgs& a_in, int groups, int mode, int var, int mj, int nw_force, hipStream_t s, int ring32) {
    ConvArgs a = a_in;
    if (a.nterms) a.staged_out = g_f32_direct_epilogue ? 0 : 1;      // plane launches: the LDS-staged fp32 epilogue of the 256 x 256 tile
    const bool w4 = ring32 == 2;                       // (2: the four-wave kernel, LOFT_CONV_W4)
    const bool xfirst = ring32 == 3 || ring32 == 5;    // (3: the activations-first stream schedule, LOFT_CONV_XFIRST)
    const bool lean = ring32 == 4 || ring32 == 5;      // (4: LOFT_CONV_LEAN, 5: LOFT_CONV_LEANX = LEAN + XFIRST)
    if (w4 || xfirst || lean) ring32 = 0;
    if ((xfirst || lean) && !(mode == 1 && var == 0 && mj == 4 && nw_force == 0 && a.Cout % 256 == 0 && !a.nterms))
        return (int)hipErrorInvalidValue;
    if (lean && a.tap_major) return (int)hipErrorInvalidValue;
    if (w4 && !(mode == 1 && var == 0 && mj == 4 && nw_force == 0 && a.Cout % 256 == 0 && !a.tap_major && !a.krot && !a.nterms))
        return (int)hipErrorInvalidValue;
    if (ring32 && !(mode == 1 && var == 0 && mj == 4 && nw_force == 0 && a.Cout % 256 == 0 && !a.tap_major && !a.krot && !a.nterms))
        return (int)hipErrorInvalidValue;
    if (mj != 4 && !((mj == 2 || mj == 1) && mode == 1 && var == 0)) return (int)hipErrorInvalidValue;
    if (mode == 2 && (a.Cout % 256 || nw_force == 1)) return (int)hipErrorInvalidValue;
    const int nw = nw_force == 1 ? 1 : (a.Cout % 256 == 0 ? 2 : 1);       // 128-cout tiles: Cout = 128 (mod 256), or by choice with 64-pixel tiles
    if (nw == 1 && !(mode == 1 && var == 0 && (mj == 4 || mj == 1) && a.Cout % 128 == 0)) return (int)hipErrorInvalidValue;
    if (a.nterms && !(mode == 1 && var == 0 && a.nterms <= CONV_MAX_TERMS && a.nterms * a.T <= 64)) return (int)hipErrorInvalidValue;
    if (a.par_n && !(mode == 1 && var == 0 && mj == 4 && nw == 2 && a.Cout == 256 && !w4 && !ring32 && !a.nterms && a.T == 1 && a.os == 2 &&
                     !a.residual && !a.mask && a.staged_out))
        return (int)hipErrorInvalidValue;
    dim3 grid(loft_cdiv(a.M, 64 * mj), (a.par_n ? 4 : 1) * (a.Cout / (128 * nw)), groups);
    fastdiv_setup(grid.x * grid.y, &a.gxy_mul, &a.gxy_sh);
    fastdiv_setup(grid.x, &a.gx_mul, &a.gx_sh);
    fastdiv_setup(grid.y, &a.gy_mul, &a.gy_sh);
    a.pointwise = a.T == 1 && a.dy[0] == 0 && a.dx[0] == 0 && a.ss == 1 && !a.pixmajor && a.IH == a.OH && a.IW == a.OW;
    a.dy_pk = a.dx_pk = a.wt_pk = 0ull;
    a.pk_ok = 1;
    for (int t = 0; t < a.T; ++t) {
        if (a.dy[t] < -8 || a.dy[t] > 7 || a.dx[t] < -8 || a.dx[t] > 7 || a.wt[t] < 0 || a.wt[t] > 15) { a.pk_ok = 0; break; }
        a.dy_pk |= (unsigned long long)(a.dy[t] + 8) << (4 * t);
        a.dx_pk |= (unsigned long long)(a.dx[t] + 8) << (4 * t);
        a.wt_pk |= (unsigned long long)a.wt[t] << (4 * t);
    }
#define PIPE_LAUNCH(M_, V_) hipLaunchKernelGGL((conv_tap_pipe_kernel<M_, V_>), grid, dim3(512), 0, s, a)
    if (a.nterms) {                   // operand planes: the stream schedule's five tile shapes with the fp32 epilogue
        if (nw == 1 && mj == 1) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 1, 1, true>), grid, dim3(512), 0, s, a);
        else if (nw == 1) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 1, true>), grid, dim3(512), 0, s, a);
        else if (mj == 2) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 2, 2, true>), grid, dim3(512), 0, s, a);
        else if (mj == 1) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 1, 2, true>), grid, dim3(512), 0, s, a);
        else if (g_stream_form >= 2 && !a.tap_major)        // (the lean instruction stream, as for the 16-bit launches: same sums, same order)
            hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, true, false, false, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, true>), grid, dim3(512), 0, s, a);
    } else if (xfirst && lean) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, true, true>), grid, dim3(512), 0, s, a);
    } else if (lean) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, false, true>), grid, dim3(512), 0, s, a);
    } else if (xfirst) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, true>), grid, dim3(512), 0, s, a);
    } else if (w4) {
        hipLaunchKernelGGL(conv_tap_w4_kernel<0>, grid, dim3(256), 0, s, a);
    } else if (ring32) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, true>), grid, dim3(512), 0, s, a);
    } else if (nw == 1 && mj == 1) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 1, 1>), grid, dim3(512), 0, s, a);
    } else if (nw == 1) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 1>), grid, dim3(512), 0, s, a);
    } else if (mode == 1 && mj == 2) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 2>), grid, dim3(512), 0, s, a);
    } else if (mode == 1 && mj == 1) {
        hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 1>), grid, dim3(512), 0, s, a);
    } else if (mode == 2) {
        if (var != 0) return (int)hipErrorInvalidValue;
        PIPE_LAUNCH(2, 0);
    } else if (mode == 1 && var == 0 && g_stream_form != 0 && !(a.tap_major && (g_stream_form == 2 || g_stream_form == 3))) {
        // the process-wide default form of the two-stage 256 x 256 stream schedule (loft_conv_stream_form: same-box A/B of a whole step)
        if (g_stream_form == 1) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, true>), grid, dim3(512), 0, s, a);
        else if (g_stream_form == 2) hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, false, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv_tap_pipe_kernel<1, 0, 4, 2, false, false, true, true>), grid, dim3(512), 0, s, a);
    } else if (mode == 1) {
        switch (var) {
        case 0: PIPE_LAUNCH(1, 0); break;
        case 1: PIPE_LAUNCH(1, 1); break;
        case 2: PIPE_LAUNCH(1, 2); break;      // direct (unstaged) epilogue
        case 3: PIPE_LAUNCH(1, 3); break;
        case 4: PIPE_LAUNCH(1, 4); break;
        case 8: PIPE_LAUNCH(1, 8); break;
        case 10: PIPE_LAUNCH(1, 10); break;
        case 12: PIPE_LAUNCH(1, 12); break;
        case 14: PIPE_LAUNCH(1, 14); break;
        case 9: PIPE_LAUNCH(1, 9); break;       // TRACE + no copies
        case 15: PIPE_LAUNCH(1, 15); break;     // TRACE + halo-volume copies
        default: return (int)hipErrorInvalidValue;
        }
    } else {
        switch (var) {
        case 0: PIPE_LAUNCH(0, 0); break;
        case 1: PIPE_LAUNCH(0, 1); break;
        case 4: PIPE_LAUNCH(0, 4); break;
        case 5: PIPE_LAUNCH(0, 5); break;
        case 8: PIPE_LAUNCH(0, 8); break;
        case 10: PIPE_LAUNCH(0, 10); break;
        case 12: PIPE_LAUNCH(0, 12); break;
        default: return (int)hipErrorInvalidValue;
        }
    }
#undef PIPE_LAUNCH
    LOFT_LAUNCH_CHECK();
    return 0;
}
