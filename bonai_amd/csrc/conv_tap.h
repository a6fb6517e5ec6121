// conv_tap.h -- shared pieces of the NHWC "tap" convolution kernels (conv_mfma.hip, conv_pipe.hip): argument block, LDS swizzle,
// XCD-aware workgroup order and the fused epilogue.  See conv_mfma.hip for the contract.
#pragma once
#include "loft_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

#define CONV_MAX_TAPS 16
#define CONV_MAX_TERMS 8      /* operand-plane products per launch (loft_conv_tap_planes / loft_conv_wgrad_planes) */
#define WGRAD_MAX_VGROUPS 32 /* groups x terms of a plane weight-gradient launch */
#define BK 64

struct ConvArgs {
    // ---- block 1 (44 dwords): every scalar the pipelined kernels' set-up reads, CONTIGUOUS and in first-use order so that they
    // arrive in a few wide s_loads issued together at kernel entry (round 4: scattered over the struct they came in ~25 dependent
    // scalar-memory round trips, most of the 8.7k-cycle set-up of a 147k-cycle tile)
    const bf16_t* src;
    const bf16_t* wgt;
    const bf16_t* zero_page;
    void* out;
    long src_gs, wgt_gs;
    unsigned gxy_mul, gxy_sh, gx_mul, gx_sh, gy_mul, gy_sh;   // conv_pipe.hip: exact division by grid.x * grid.y, grid.x, grid.y
    int nfast;               // tile order: channel tiles of one pixel tile adjacent (the pixel tile is read from HBM once)
    int pixmajor;            // FAST kernels on small RoI maps: tile rows enumerate (pixel, RoI) instead of (RoI, pixel) -- see below
    int pointwise;   // conv_pipe.hip: one tap at offset 0, unit strides, input map = output map: row m reads input pixel m
    int T;
    int B, IH, IW, Cin, Cout;
    int OH, OW;
    int M;
    int ss;
    // conv_pipe.hip, pixel-major rows in RoI BLOCKS: row m -> segment m / pm_S (pm_S rows = the RoIs of one block at one pixel
    // position), block = segment / pm_P, position = segment % pm_P, RoI = block * pm_S + m % pm_S (rows with RoI >= B are
    // padding: pm_S = ceil(B / number of blocks) >= 256).  The 49 / 196 positions of a block's RoIs are then CONSECUTIVE tiles --
    // one XCD at a time works on one block, whose rows (1.4 MB per 64-channel chunk) stay in its L2 while the taps and the
    // neighbouring positions re-read them; with position-major order over all B RoIs each tile's 9 taps touched rows that no
    // concurrently running tile shared (fetch 4.1x the input, tools/pmc_traffic_shapes.sh).  M counts the padded rows.
    int pm_S, pm_P;
    unsigned pms_mul, pms_sh, pmp_mul, pmp_sh;
    unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;   // exact division by OH*OW, OW via multiply-high (host-computed)
    int tap_major;   // conv_pipe.hip: K order (tap, chunk) instead of (chunk, tap); see the kernel
    int krot;        // conv_pipe.hip, chunk-major order: workgroup V starts at channel chunk (V % krot_n) and wraps -- at any instant the
                     // workgroups of the chip read DIFFERENT weight tiles instead of all 256 CUs fetching the same 32 KiB
    // conv_pipe.hip: the tap tables as 4-bit fields of three 64-bit scalars (tap t: bits 4t..4t+3; dy + 8, dx + 8, wt), valid when
    // pk_ok (every |dy|, |dx| <= 7 -- always, for the 3x3 / strided-parity tap sets of this model): lane t unpacks its entry with
    // two VALU ops instead of loading it from the kernarg segment (a vector-memory round trip in the middle of the set-up)
    unsigned long long dy_pk, dx_pk, wt_pk;
    int pk_ok;
    // ---- block 2: the epilogue's scalars
    const float* bias;
    const bf16_t* residual;
    const bf16_t* mask;      // optional: zero the result where mask <= 0 (ReLU backward of the tensor this gradient is for)
    long out_gs, bias_gs;
    int OHf, OWf;
    int os, oo_y, oo_x;
    int relu, out_f32, accumulate;
    int staged_out;          // 128x128 kernel, dense bf16 output: collect the tile in LDS and store it row-contiguously
    unsigned b_mul, b_sh;    // exact division by B
    void* trace;             // conv_pipe.hip TRACE variants only: device buffer for barrier time stamps
    // ---- block 3: per-tap tables (read by lanes, once)
    int dy[CONV_MAX_TAPS], dx[CONV_MAX_TAPS], wt[CONV_MAX_TAPS];
    // ---- block 4: operand PLANES (loft_conv_tap_planes, conv_pipe.hip PL instances; nterms == 0: a plain launch).  The fp32 parity
    // mode hands both operands over as NP 16-bit planes of one fp32 tensor each (x = sum_p plane_p / scale, loft_split_planes_f32)
    // and the product as `nterms` (activation plane, weight plane) pairs: term p reads activation plane xoff[p] / weight plane
    // woff[p] (ELEMENT offsets of the plane inside src / wgt).  The kernel runs the terms as extra taps of ONE K loop -- every
    // term accumulates into the same fp32 accumulators -- and its epilogue is fp32: residual / mask / out are float tensors, the
    // accumulators are scaled by 1 / (scale_x * scale_w) derived from the two absmax scalars (null: unscaled planes).
    int nterms;
    int xoff[CONV_MAX_TERMS], woff[CONV_MAX_TERMS];
    const float* amax_x;
    const float* amax_w;
    float* amax_out;         // optional (plane launches): max |stored output| is atomically folded into this PRE-ZEROED word, so that
                             // the next layer's plane split of `out` needs no absmax pass of its own
    // ---- block 5: a NARROW 1x1 HEAD on the output, computed from the staged bf16 tile before it leaves LDS (conv_pipe.hip, staged
    // epilogue, 256-cout tiles; loft_conv_tap_bf16_head).  head_out[pixel][n] = head_b[n] + sum_c bf16(out[pixel][c]) * head_w[n][c]
    // for n < head_c4 <= 32: the RPN's objectness + delta convs behind its 3x3 conv, the mask logits behind the deconvolution --
    // the wide map is written once and not read back by a second launch.  head_w: bf16 [head_c4][256].
    const bf16_t* head_w;
    const float* head_b;     // fp32 [head_c4]
    float* head_out;         // fp32 [pixels of the FULL output map][head_c4]
    int head_c4;
    int par_n;               // conv_pipe.hip stream kernel (loft_deconv2x2_bf16): the launch has FOUR N tiles of Cout = 256 channels each,
                             // N tile p = tap p of a 2x2 / stride-2 deconvolution (weight rows [256 p, 256 p + 256) of wgt), stored at
                             // output offset (oo_y + (p >> 1), oo_x + (p & 1)) with os = 2: one launch, the input tile read once
};

// Power-of-two scale of a plane split (loft_split_planes_f32 with an absmax scalar; the binary16 build): the tensor's absmax lands
// in [2^14, 2^15) -- the high plane cannot overflow binary16 and an element 2^-18 below the absmax still keeps all 22 bits of its
// two planes.  amax == 0, inf or NaN: scale 1.  Exact (a power of two), so is its inverse.
__host__ __device__ __forceinline__ float planes_scale_of(float amax, bool inverse) {
    union { float f; uint32_t u; } c;
    c.f = amax;
    const int e = (int)((c.u >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;
    int se = 14 - (e - 127);                       // scale = 2^se
    if (inverse) se = -se;
    se = se < -126 ? -126 : (se > 127 ? 127 : se);
    c.u = (uint32_t)(se + 127) << 23;
    return c.f;
}

// A kernarg scalar made OPAQUE to the compiler at this point: hipcc treats ConvArgs fields as rematerialisable loads and re-fetches
// them from scalar memory (s_load + s_waitcnt lgkmcnt(0), ~100-200 cycles each) inside unrolled row loops instead of keeping them
// in SGPRs -- the epilogues' row stores ran 6-9 such round trips apiece (round 4, ISA of conv_tap_kernel / conv_tap_pipe_kernel).
// After LOFT_KEEP_S(x) the value lives in a register the compiler cannot re-derive from memory.
#define LOFT_KEEP_S(x) asm volatile("" : "+s"(x))

// n / d for 0 <= n < 2^31 with (mul, sh) = fastdiv_setup(d): q = (umulhi(n, mul) + n) >> sh  (Granlund-Montgomery)
__device__ __forceinline__ int fastdiv(int n, unsigned mul, unsigned sh) {
    return (int)(((unsigned long long)__umulhi((unsigned)n, mul) + (unsigned)n) >> sh);
}
static inline void fastdiv_setup(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    *mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *sh = l;
}

__device__ __forceinline__ int swz(int row, int q) { return q ^ ((row >> 1) & 7); }

// XCD-aware workgroup order (speed only, never correctness): hardware places linear workgroup L on XCD L % 8, each with
// a private L2.  Re-label so that every XCD works on ONE contiguous range of the logical tile order -- neighbouring
// tiles (3x3 halo rows of adjacent pixel tiles; all taps / channel tiles of one pixel range in wgrad) then share an L2
// instead of being fetched eight times.  Bijective for any workgroup count.
__device__ __forceinline__ int xcd_remap(int L, int N) {
    const int xcd = L & 7, q = N >> 3, r = N & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (L >> 3);
}

// Shared epilogue of the tap-conv kernels: bias (= folded BN shift) + residual + ReLU + ReLU-backward mask + bf16/fp32 store.
// res_t / mask_t (optional): the 128x128 bf16 tiles of a.residual / a.mask staged in LDS by conv_stage_tile() with coalesced
// 16-byte global->LDS copies (row r, logical 16-byte chunk c at r*256 + ((c ^ (r & 15)) * 16)); the per-lane 8-byte reads then
// hit LDS instead of scattering 8-byte loads over 32 different 64-byte sectors of HBM/L2 per instruction.
template <int NT, int MT, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NT][MT], int g, int m0, int n0, int wm, int wn,
                                              int frow, int fq, int ohw, const char* res_t = nullptr,
                                              const char* mask_t = nullptr, char* out_t = nullptr, bool pixmajor = false) {
    // ---- epilogue: lane holds, per (i,j) tile, pixel m = ..+(lane&31) and 4x4 consecutive channels.
    // (Measured alternatives, both slower on MI355X: swapping the MFMA operands so lanes run along channels and storing
    //  2-byte scalars -- 5x slower, sub-dword stores do not coalesce; the same with a DPP pair exchange and dword stores in
    //  64-byte runs -- 8 % slower end to end: store width per lane matters more than run contiguity, L2 merges the lines.)
    const float* bias = a.bias ? a.bias + (long)g * a.bias_gs : nullptr;
    const long out_g = (long)g * a.out_gs;
    // every scalar of the row / store loops in registers the compiler cannot re-fetch from the kernarg segment (LOFT_KEEP_S)
    int aM = a.M, aB = a.B, aOW = a.OW, aOHf = a.OHf, aOWf = a.OWf, aos = a.os, aoy = a.oo_y, aox = a.oo_x, aCout = a.Cout;
    int arelu = a.relu, af32 = a.out_f32, aacc = a.accumulate;
    const bf16_t* ares = a.residual;
    const bf16_t* amask = a.mask;
    void* aout = a.out;
    LOFT_KEEP_S(aM); LOFT_KEEP_S(aB); LOFT_KEEP_S(aOW); LOFT_KEEP_S(aOHf); LOFT_KEEP_S(aOWf); LOFT_KEEP_S(aos); LOFT_KEEP_S(aoy);
    LOFT_KEEP_S(aox); LOFT_KEEP_S(aCout); LOFT_KEEP_S(arelu); LOFT_KEEP_S(af32); LOFT_KEEP_S(aacc); LOFT_KEEP_S(ares);
    LOFT_KEEP_S(amask); LOFT_KEEP_S(aout);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * WM + j * 32 + frow;
        if (m >= aM) continue;
        int b, rem;
        if (pixmajor) { rem = m / aB; b = m - rem * aB; } else { b = m / ohw; rem = m - b * ohw; }
        const int oy = rem / aOW, ox = rem - oy * aOW;
        const long opix = ((long)b * aOHf + oy * aos + aoy) * aOWf + ox * aos + aox;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * WN + i * 32 + 8 * gq + 4 * fq;
                if (n >= aCout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][gq * 4 + e];
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                const long o = out_g + opix * aCout + n;
                const int trow = wm * WM + j * 32 + frow;                      // position inside a staged 128x128 tile
                const int tcol = (((wn * WN + i * 32 + 8 * gq) >> 3) ^ (trow & 15)) * 16 + 8 * fq;
                if (ares) {
                    float rv[4];
                    if (res_t) ld4(reinterpret_cast<const bf16_t*>(res_t + trow * 256 + tcol), rv);
                    else ld4(ares + o, rv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                if (arelu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (amask) {
                    float mv[4];
                    if (mask_t) ld4(reinterpret_cast<const bf16_t*>(mask_t + trow * 256 + tcol), mv);
                    else ld4(amask + o, mv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
                }
                if (out_t) {              // bf16 tile collected in LDS (same swizzle), written out by conv_unstage_tile
                    st4(reinterpret_cast<bf16_t*>(out_t + trow * 256 + tcol), v);
                } else if (af32) {
                    float* op = reinterpret_cast<float*>(aout) + o;
                    if (aacc) {
                        float ov[4];
                        ld4(op, ov);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ov[e];
                    }
                    st4(op, v);
                } else {
                    st4(reinterpret_cast<bf16_t*>(aout) + o, v);
                }
            }
        }
    }
}

// ---- weight gradient (conv_mfma.hip conv_wgrad_kernel, conv_wgrad_pipe.hip): argument block and LDS swizzle
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define WGRAD_PM_MAX_BLOCKS 256
struct WgradArgs {
    const bf16_t* g;
    const bf16_t* x;
    float* dw;
    const bf16_t* zero_page;
    int B, GH, GW, Cout, XH, XW, Cin, OH, OW;
    int gos, ss, T;
    int goy[CONV_MAX_TAPS], gox[CONV_MAX_TAPS], dy[CONV_MAX_TAPS], dx[CONV_MAX_TAPS], wt[CONV_MAX_TAPS];
    long g_gs, x_gs, dw_gs;
    int M, pix_per_split, ctiles;
    unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;   // exact division by OH*OW and OW via multiply-high (host-computed)
    float* db;      // optional bias gradient db[g][n] += sum_pixels G (fp32 atomics), fused: see db_tap
    int db_tap;     // tap whose X gather is never out of bounds (its G rows are complete); -2 = every tap; -1 = off
    // PM form (RoI maps of 7x7 / 14x14 pixels, hundreds of RoIs; stride 1): rows whose tap leaves the map (18 % of a 7x7 map's
    // 3x3 taps, 9 % of a 14x14 one) are never staged.  K-split s of tap t reduces over RoIs [s * pm_pps[t], (s + 1) * pm_pps[t])
    // at ALL positions of the tap's valid rectangle, local row j -> (position j / pm_pps[t], RoI s * pm_pps[t] + j % pm_pps[t]),
    // pm_rows[t] = positions * pm_pps[t] rows per split (RoIs >= B are padding rows); pm_pps[t] is chosen per tap so that all
    // workgroups run the same number of K-steps (more splits for taps with more valid positions).  With the RoI range outermost,
    // the splits of the nine taps that run side by side on one XCD (pm_tap / pm_split below) walk the SAME RoIs, a few positions
    // apart: a G / X row is re-read by another tap within ~1 MB of traffic and comes from that XCD's L2 (position-major order over
    // all B RoIs put 3.6 MB between two taps' reads of one row: 5x the operand bytes fetched per launch).
    // Workgroup j of a group serves tap pm_tap[j]; tap t has pm_blk0[t+1] - pm_blk0[t] splits.
    int pm_blk0[CONV_MAX_TAPS + 1], pm_pps[CONV_MAX_TAPS], pm_rows[CONV_MAX_TAPS];
    unsigned pm_pps_mul[CONV_MAX_TAPS], pm_pps_sh[CONV_MAX_TAPS];
    // Block order of the PM form: block j (position in the launch, after the XCD-aware remap) -> tap pm_tap[j], K-split
    // pm_split[j] of that tap, sorted by the split's relative position in its tap's rows.  The taps of one pixel range then sit
    // next to each other -- on one XCD, which fetches the G / X rows they share into its L2 once instead of once per tap.
    unsigned char pm_tap[WGRAD_PM_MAX_BLOCKS], pm_split[WGRAD_PM_MAX_BLOCKS];
    // Split-K combination.  partial == 0: fp32 atomics into dw (the caller zeroed it).  partial == 1: workgroup (tap, tile, split s)
    // STORES its tile into slot s of dw viewed as [group][nslots][wtap][Cout][Cin] (split_stride = wtaps * Cout * Cin elements),
    // nothing is pre-zeroed and the consumer (loft_fold_unpack_bwd_multi) sums the slots while it reads: an fp32 atomic costs the
    // L2 one channel-cycle per 4 bytes (measured 0.29 T atomics/s over the chip, tools/probes/atomic_scope.hip), a store 1/16 of
    // that.  The last split of a tap with fewer than nslots splits also zero-fills that tap's remaining slots.
    int partial, nslots;
    long split_stride;
    int pm_inc_ok;  // conv_wgrad_pipe.hip: G and X hold fewer than 2^31 elements each (32-bit running offsets in the incremental PM decode)
    int pm_y0[CONV_MAX_TAPS], pm_x0[CONV_MAX_TAPS], pm_rw[CONV_MAX_TAPS];
    unsigned pm_rw_mul[CONV_MAX_TAPS], pm_rw_sh[CONV_MAX_TAPS], b_mul, b_sh;
    // Operand planes (loft_conv_wgrad_planes; nvg == 0: a plain launch).  The launch's "groups" are nvg VIRTUAL groups = (real
    // group, term): virtual group v reads the G plane at element offset vg_g[v], the X plane at vg_x[v] and adds into the dW of
    // its real group at vg_dw[v] -- the terms of one product meet in dW through the split-K atomics that are there anyway, and the
    // split count shrinks by the number of terms (same workgroups, same atomics as one 16-bit launch).  amax_*: see ConvArgs.
    int nvg;
    long vg_g[WGRAD_MAX_VGROUPS], vg_x[WGRAD_MAX_VGROUPS], vg_dw[WGRAD_MAX_VGROUPS];
    // bias gradient of a plane launch (round 6): virtual group v adds the column sums of ITS G plane into db[vg_db[v] + n]
    // (vg_db[v] < 0: this term does not -- every G plane is summed by exactly one term, the one that pairs it with X plane 0),
    // un-scaled by 1 / scale_g alone; replaces a separate column-sum pass over the fp32 gradient map
    int vg_db[WGRAD_MAX_VGROUPS];
    const float* amax_g;
    const float* amax_x;
};
// group base offsets of a weight-gradient workgroup (kernel top level only: see the note on closures in conv_wgrad_pipe.hip)
#define WGRAD_G_OFF(a, grp) ((a).nvg ? (a).vg_g[grp] : (long)(grp) * (a).g_gs)
#define WGRAD_X_OFF(a, grp) ((a).nvg ? (a).vg_x[grp] : (long)(grp) * (a).x_gs)
#define WGRAD_DW_OFF(a, grp) ((a).nvg ? (a).vg_dw[grp] : (long)(grp) * (a).dw_gs)
#define WGRAD_OUT_SCALE(a) ((a).amax_g ? planes_scale_of(*(a).amax_g, true) * planes_scale_of(*(a).amax_x, true) : 1.f)
#define WGRAD_DB_ON(a, grp) ((a).db != nullptr && (!(a).nvg || (a).vg_db[grp] >= 0))
#define WGRAD_DB_PTR(a, grp) ((a).db + ((a).nvg ? (long)(a).vg_db[grp] : (long)(grp) * (a).Cout))
#define WGRAD_DB_SCALE(a) (((a).nvg && (a).amax_g) ? planes_scale_of(*(a).amax_g, true) : 1.f)

__device__ __forceinline__ int wswz(int row, int q) { return q ^ ((row & 3) << 2); }

