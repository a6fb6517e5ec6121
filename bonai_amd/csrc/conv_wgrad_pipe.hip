// conv_wgrad_pipe.hip -- weight gradient of the tap convolution (contract: conv_mfma.hip "Weight gradient") for 256 x 256 tiles as
// ONE software-pipelined instruction stream per wave -- the weight-gradient twin of conv_pipe.hip's stream schedule.
//
//   dW[wt[t]][n][c] += sum_m  G[b, oy*gos+goy[t], ox*gos+gox[t], n] * X[b, oy*ss+dy[t], ox*ss+dx[t], c]
//
// Reference call sites: the autograd of every nn.Conv2d / nn.Linear with Cout, Cin multiples of 256 on the LOFT path (FOA branches
// offset_head_expand_feature.py:134-161, mask head fcn_mask_head.py:118-126, FPN / RPN 3x3 fpn.py:170-199, rpn_head.py:38-44, FCs).
//
// What was wrong with conv_wgrad_kernel<256,8> (ISA of round 1's build): (1) its fragments come from
// __builtin_amdgcn_ds_read_tr16_b64, and hipcc fences that builtin behind EVERY outstanding global->LDS copy: an
// `s_waitcnt vmcnt(0)` sat between stage() and the first fragment read, so the "prefetch" of K-step s+1 was waited for before
// K-step s was computed; (2) all eight waves issued their eight copies, each with a full pixel decode, in lockstep in front of
// the MFMAs.  Here: the transposing reads are inline asm (ds_read_b64_tr_b16 with compile-time offsets from six per-lane base
// registers, waited for by hand), fragments are double-buffered one 16-pixel sub-step ahead also across K-steps, one barrier
// per K-step, copies are spread over two sub-steps, each pixel is decoded once per K-step for both operands.
//
// LDS (128 KiB, one array): [G buf0][G buf1][X buf0][X buf1], 32 KiB each = 64 pixel rows x 512 B (256 channels); 16-byte chunk
// q of pixel row r at q ^ ((r & 3) << 2).  Waves: 2 (n) x 4 (c); wave tile 128 (n) x 64 (c): 4 G fragments + 2 X fragments and
// 8 MFMAs per sub-step.  Schedule per K-step s (buffer B = s & 1):
//   ks0: MFMA fa | read F(s,1) -> fb | issue X(s+2)... see kernel body for the exact windows
// Roofline: MFMA; algorithmic work 2*M*Cout*Cin*T FLOP per launch.
#include "conv_tap.h"
#include <type_traits>
#include "../../include/loft_hip.h"

namespace {
constexpr int WG_OFF = 0, WX_OFF = 65536, WBUF = 32768, WLDS = 131072;
}
#define WSB() __builtin_amdgcn_sched_barrier(0)

// one transposing read: 4 consecutive pixels (k) of one channel for each lane of a 16-lane group -- half of an MFMA operand
#define TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))

// MODE 0: generic taps (per-row decode); 1: PM (RoI maps, WgradArgs); 2: DENSE and 3: SAME -- the decode-free row addressing of
// conv_wgrad_ring_kernel below (1x1 layers / FCs: reduction row m is pixel m of both operands; stride-1 same-size taps: G row =
// pixel m, X row = pixel m + dy * W + dx inside the map, (oy, ox) of a thread's four rows carried from K-step to K-step).
template <int MODE>
__global__ __launch_bounds__(512) void conv_wgrad_stream_kernel(const WgradArgs a) {
    // (MODE 4: PM with the per-row decode -- splits of fewer than 64 RoIs, or tensors too large for 32-bit running offsets)
    constexpr bool PM = MODE == 1 || MODE == 4, PMINC = MODE == 1, DENSE = MODE == 2, SAME = MODE == 3;
    __shared__ __attribute__((aligned(16))) char lds[WLDS];
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
    const int n0 = nt * 256, c0 = ct * 256;
    if (mbeg >= mend) return;
    const bf16_t* G = a.g + WGRAD_G_OFF(a, grp);
    const bf16_t* X = a.x + WGRAD_X_OFF(a, grp);
    const int ohw = a.OH * a.OW;
    const int goy = a.goy[t], gox = a.gox[t], dy = a.dy[t], dx = a.dx[t];
    const int pm_y0 = PM ? a.pm_y0[t] : 0, pm_x0 = PM ? a.pm_x0[t] : 0, pm_rw = PM ? a.pm_rw[t] : 1;
    const unsigned pm_mul = PM ? a.pm_rw_mul[t] : 0u, pm_sh = PM ? a.pm_rw_sh[t] : 0u;
    const int pm_rb = PM ? a.pm_pps[t] : 1, pm_b0 = PM ? (int)a.pm_split[bz] * pm_rb : 0;
    const unsigned rb_mul = PM ? a.pm_pps_mul[t] : 0u, rb_sh = PM ? a.pm_pps_sh[t] : 0u;

    // ---- staging: thread -> pixel rows i*16 + wave*2 + (lane>>5), i = 0..3, 16-byte chunk lane & 31 (swizzled on the source)
    const int srow = wave * 2 + (lane >> 5);
    const int schunk = (lane & 31) ^ ((srow & 3) << 2);           // rows 16 apart share (row & 3)
    const bf16_t* px_next[4];                                      // X pointers of the K-step whose G rows were issued last
    int s_oy[4] = {0, 0, 0, 0}, s_ox[4] = {0, 0, 0, 0};         // SAME: map position of row i at its next decode
    const int xshift = dy * a.XW + dx;
    if constexpr (SAME) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbeg + i * 16 + srow;
            const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
            s_oy[i] = fastdiv(rem, a.ow_mul, a.ow_sh);
            s_ox[i] = rem - s_oy[i] * a.OW;
        }
    }
    // PM, incremental (round 4): the ISA of the K loop showed the per-row decode below -- two multiply-high divisions and two chains
    // of 64-bit multiplies per staged row, 32 quarter-rate v_mul_lo_u32 + 8 v_mul_hi_u32 + 16 64-bit multiply-adds per wave and
    // K-step -- as ~1400 cycles of VALU issue per wave against 1152 of MFMA, for BOTH waves of a SIMD: the RoI-map launches were
    // VALU-bound on their own addressing.  A thread's row i advances by 64 reduction rows per K-step; with >= 64 RoIs per split
    // that is 64 RoIs further at the same position or, once, a wrap to the next position of the tap's rectangle (next column, or
    // first column of the next map row): its G / X element offsets move by one of three wave-uniform increments.  State per row:
    // the reduction index, the RoI offset inside the split, the column inside the rectangle, the two offsets.  No multiplies.
    // (32-bit element offsets: the host enables the incremental form only when both tensors have fewer than 2^31 elements)
    const int pm_B = a.B;
    const bf16_t* zpage = a.zero_page;
    constexpr bool pm_inc = PMINC;          // (the host launches MODE 1 only when every split holds >= 64 RoIs and pm_inc_ok)
    int pm_r[4] = {0, 0, 0, 0}, pm_rx[4] = {0, 0, 0, 0}, pm_og[4] = {0, 0, 0, 0}, pm_ox[4] = {0, 0, 0, 0};
    int ig_step = 0, ig_col = 0, ig_row = 0, ix_step = 0, ix_col = 0, ix_row = 0, v_rb = 0, v_rw = 1;
    if constexpr (PMINC) {
        {
            const int sg = a.GH * a.GW * a.Cout, sx = a.XH * a.XW * a.Cin;
            // increments of a row's element offsets per K-step: +64 RoIs; on a position wrap additionally (-RoIs of the split, next
            // column); on a wrap at the rectangle's last column additionally (first column of the next map row instead)
            ig_step = 64 * sg; ig_col = a.Cout - pm_rb * sg; ig_row = (a.GW - pm_rw) * a.Cout;
            ix_step = 64 * sx; ix_col = a.Cin - pm_rb * sx; ix_row = (a.XW - pm_rw) * a.Cin;
            v_rb = pm_rb; v_rw = pm_rw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mbeg + i * 16 + srow;
                const int p = fastdiv(m, rb_mul, rb_sh), r = m - p * pm_rb;
                const int ry = fastdiv(p, pm_mul, pm_sh), rx = p - ry * pm_rw;
                const int b = pm_b0 + r, oy = pm_y0 + ry, ox = pm_x0 + rx;
                pm_r[i] = r; pm_rx[i] = rx;
                pm_og[i] = ((b * a.GH + oy + goy) * a.GW + ox + gox) * a.Cout + n0 + schunk * 8;
                pm_ox[i] = ((b * a.XH + oy + dy) * a.XW + ox + dx) * a.Cin + c0 + schunk * 8;
            }
        }
    }
    // decode pixel m of the reduction range once: G pointer (or the zero page), X pointer (or the zero page)
    auto decode = [&](int m, const bf16_t*& pg, const bf16_t*& px, int i) {
        pg = zpage; px = zpage;
        if constexpr (PMINC) {
            {                      // (row i is decoded exactly once per K-step, in step order: prologue steps 0 and 1, then s + 2)
                const bool ok = (m < mend) & (pm_b0 + pm_r[i] < pm_B);
                if (ok) { pg = G + pm_og[i]; px = X + pm_ox[i]; }
                const int r2 = pm_r[i] + 64, rx2 = pm_rx[i] + 1;
                const bool w = r2 >= v_rb, wr = w & (rx2 == v_rw);
                pm_r[i] = w ? r2 - v_rb : r2;
                pm_rx[i] = w ? (wr ? 0 : rx2) : pm_rx[i];
                // (sums of masked terms, not nested selects: hipcc turns a nested ?: on wave-uniform values into control flow and
                //  then keeps the closure's state in scratch)
                pm_og[i] += ig_step + (w ? ig_col : 0) + (wr ? ig_row : 0);
                pm_ox[i] += ix_step + (w ? ix_col : 0) + (wr ? ix_row : 0);
                return;
            }
        } else {                   // (discarded for the incremental instance: its closure must not reference the kernarg struct --
                                   //  hipcc otherwise copies all of WgradArgs to scratch and indexes the per-tap tables there)
        if constexpr (DENSE) {
            if (m < mend) {
                pg = G + (long)m * a.Cout + n0 + schunk * 8;
                px = X + (long)m * a.Cin + c0 + schunk * 8;
            }
            return;
        }
        if constexpr (SAME) {
            const int iy = s_oy[i] + dy, ix = s_ox[i] + dx;
            if ((m < mend) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                pg = G + (long)m * a.Cout + n0 + schunk * 8;
                px = X + (long)(m + xshift) * a.Cin + c0 + schunk * 8;
            }
            s_ox[i] += 64;                                          // this row's pixel at the next K-step (OW >= 64: one wrap at most)
            if (s_ox[i] >= a.OW) {
                s_ox[i] -= a.OW;
                s_oy[i] = s_oy[i] + 1 == a.OH ? 0 : s_oy[i] + 1;
            }
            return;
        }
        if (m < mend) {
            if constexpr (PM) {     // m = (position inside the valid rectangle of tap t) * RoIs-per-split + RoI; stride 1, in bounds
                const int p = fastdiv(m, rb_mul, rb_sh), b = pm_b0 + (m - p * pm_rb);
                const int ry = fastdiv(p, pm_mul, pm_sh), rx = p - ry * pm_rw;
                const int oy = pm_y0 + ry, ox = pm_x0 + rx;
                if (b < a.B) {
                    pg = G + ((long)(b * a.GH + oy + goy) * a.GW + ox + gox) * a.Cout + n0 + schunk * 8;
                    px = X + ((long)(b * a.XH + oy + dy) * a.XW + ox + dx) * a.Cin + c0 + schunk * 8;
                }
            } else {
                const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                const int gy = oy * a.gos + goy, gx = ox * a.gos + gox;
                const int iy = oy * a.ss + dy, ix = ox * a.ss + dx;
                // a product with a zero operand contributes nothing: either side may carry the zero
                if ((gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                    pg = G + ((long)(b * a.GH + gy) * a.GW + gx) * a.Cout + n0 + schunk * 8;
                    px = X + ((long)(b * a.XH + iy) * a.XW + ix) * a.Cin + c0 + schunk * 8;
                }
            }
        }
        }
    };
    // issue the G rows i0, i0+1 of K-step `step` into buffer B and remember their X pointers
    auto issue_g = [&](int step, auto bufc, auto i0c) {
        constexpr int B = decltype(bufc)::value, I0 = decltype(i0c)::value;
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i) {
            const bf16_t* pg;
            decode(mbeg + step * 64 + i * 16 + srow, pg, px_next[i], i);
            __builtin_amdgcn_global_load_lds((gptr_t)pg, (lds_ptr_t)(lds + WG_OFF + B * WBUF + (i * 16 + wave * 2) * 512), 16, 0, 0);
        }
    };
    auto issue_x = [&](auto bufc, auto i0c) {
        constexpr int B = decltype(bufc)::value, I0 = decltype(i0c)::value;
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)px_next[i], (lds_ptr_t)(lds + WX_OFF + B * WBUF + (i * 16 + wave * 2) * 512), 16, 0, 0);
    };
    using c0_t = std::integral_constant<int, 0>;
    using c1_t = std::integral_constant<int, 1>;
    using c2_t = std::integral_constant<int, 2>;
    using c3_t = std::integral_constant<int, 3>;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wn = wave >> 2, wc = wave & 3;
    // bias gradient rides along on the blocks of the first channel tile (see conv_wgrad_kernel): one extra MFMA per sub-step
    // against an all-ones operand; wave (wn, wc) takes n-block wn*4 + wc
    const bool do_db = WGRAD_DB_ON(a, grp) && ct == 0 && (a.db_tap == -2 || a.db_tap == t);
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)LOFT_ONE16;

    // ---- fragment addressing (tr_frag of conv_mfma.hip with everything lane-constant folded): lane l -> channel
    // col0 + 16*((l>>4)&1) + 4*(l&3) + [0,4) of pixel rows ks*16 + 8*(l>>5) + ((l&15)>>2) (+4 for the second half); the swizzle
    // term of a row is ((row & 3) << 2) = (((l&15)>>2) & 3) << 2 for both halves and every ks.
    const int il = lane & 15, gl = lane >> 4;
    const int frow0 = 8 * (gl >> 1) + (il >> 2);
    const int xq = (il >> 2) & 3;
    int gaddr[4], xaddr[2];                             // LDS byte addresses for ks = 0, first half, buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = wn * 128 + i * 32 + 16 * (gl & 1) + (il & 3) * 4;
        gaddr[i] = (int)(size_t)(lds + WG_OFF) + frow0 * 512 + (((col >> 3) ^ (xq << 2)) << 4) + (col & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wc * 64 + j * 32 + 16 * (gl & 1) + (il & 3) * 4;
        xaddr[j] = (int)(size_t)(lds + WX_OFF) + frow0 * 512 + (((col >> 3) ^ (xq << 2)) << 4) + (col & 7) * 2;
    }
    // fragment sets: 6 fragments x 2 halves of 4 bf16
    s16x4 fa[12], fb[12];
    // reads of fragment F of set (B, KS): 2 transposing reads
#define WG_READ(f, F, B_, KS_)                                                                         \
    do {                                                                                               \
        if ((F) < 4) {                                                                                 \
            TR_READ((f)[2 * (F)], gaddr[(F) < 4 ? (F) : 0], (B_) * WBUF + (KS_) * 8192);               \
            TR_READ((f)[2 * (F) + 1], gaddr[(F) < 4 ? (F) : 0], (B_) * WBUF + (KS_) * 8192 + 2048);    \
        } else {                                                                                       \
            TR_READ((f)[2 * (F)], xaddr[(F) >= 4 ? (F) - 4 : 0], (B_) * WBUF + (KS_) * 8192);          \
            TR_READ((f)[2 * (F) + 1], xaddr[(F) >= 4 ? (F) - 4 : 0], (B_) * WBUF + (KS_) * 8192 + 2048); \
        }                                                                                              \
    } while (0)
    // wait for every transposing read in flight; naming the registers keeps hipcc from touching them before the data has landed
#define WG_WAIT(f)                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                      \
                 : "+v"((f)[0]), "+v"((f)[1]), "+v"((f)[2]), "+v"((f)[3]), "+v"((f)[4]), "+v"((f)[5]), "+v"((f)[6]), "+v"((f)[7]), \
                   "+v"((f)[8]), "+v"((f)[9]), "+v"((f)[10]), "+v"((f)[11]))
    auto frag = [&](s16x4 (&f)[12], int F) {
        bf16x8 v;
        v[0] = f[2 * F][0]; v[1] = f[2 * F][1]; v[2] = f[2 * F][2]; v[3] = f[2 * F][3];
        v[4] = f[2 * F + 1][0]; v[5] = f[2 * F + 1][1]; v[6] = f[2 * F + 1][2]; v[7] = f[2 * F + 1][3];
        return v;
    };
    // the two MFMAs of G block I (against both X blocks)
    auto mm2 = [&](s16x4 (&f)[12], auto ic) {
        constexpr int I = decltype(ic)::value;
        const bf16x8 gv = frag(f, I);
        acc[I][0] = LOFT_MFMA_32x32x16(gv, frag(f, 4), acc[I][0]);
        acc[I][1] = LOFT_MFMA_32x32x16(gv, frag(f, 5), acc[I][1]);
    };
    auto mmb = [&](s16x4 (&f)[12]) {
        if (do_db) {   // static register selects (a runtime-indexed fragment array would be demoted to scratch)
            bf16x8 gsel = frag(f, 0);
            gsel = (wc == 1) ? frag(f, 1) : gsel;
            gsel = (wc == 2) ? frag(f, 2) : gsel;
            gsel = (wc == 3) ? frag(f, 3) : gsel;
            accb = LOFT_MFMA_32x32x16(gsel, ones, accb);
        }
    };

    const int nsteps = (mend - mbeg + 63) / 64;
    // ---- prologue: G(0), X(0), G(1) (+ its X pointers)
    issue_g(0, c0_t{}, c0_t{}); issue_g(0, c0_t{}, c2_t{});
    issue_x(c0_t{}, c0_t{}); issue_x(c0_t{}, c2_t{});
    if (nsteps > 1) {
        issue_g(1, c1_t{}, c0_t{}); issue_g(1, c1_t{}, c2_t{});
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WSB();
    __builtin_amdgcn_s_barrier();
    WSB();
    WG_READ(fa, 0, 0, 0); WG_READ(fa, 1, 0, 0); WG_READ(fa, 2, 0, 0); WG_READ(fa, 3, 0, 0); WG_READ(fa, 4, 0, 0); WG_READ(fa, 5, 0, 0);

    // one sub-step: 8 (+1) MFMAs on `cur`, the 12 transposing reads of `nxt` and up to two pairs of copies pinned between MFMA pairs
#define WG_SUBSTEP(cur, nxt, B_, KS_, do_read, copy0, copy1)                                  \
    do {                                                                                      \
        WG_WAIT(cur);                                                                         \
        WSB();                                                                                \
        mm2(cur, c0_t{});                                                                     \
        if (do_read) { WG_READ(nxt, 0, B_, KS_); WG_READ(nxt, 1, B_, KS_); }                  \
        WSB();                                                                                \
        mm2(cur, c1_t{});                                                                     \
        if (do_read) { WG_READ(nxt, 2, B_, KS_); WG_READ(nxt, 3, B_, KS_); }                  \
        copy0;                                                                                \
        WSB();                                                                                \
        mm2(cur, c2_t{});                                                                     \
        if (do_read) { WG_READ(nxt, 4, B_, KS_); WG_READ(nxt, 5, B_, KS_); }                  \
        WSB();                                                                                \
        mm2(cur, c3_t{});                                                                     \
        mmb(cur);                                                                             \
        copy1;                                                                                \
        WSB();                                                                                \
    } while (0)

    // K-step s in buffer B:
    //   ks0: MFMA fa | read F(s,1) -> fb | issue X(s+1) -> buffer 1-B  (its G rows went out in ks3(s-1); buffer 1-B is free since
    //        SYNC(s-1))
    //   ks1: MFMA fb | read F(s,2) -> fa        ks2: MFMA fa | read F(s,3) -> fb
    //   SYNC(s): vmcnt(0) (G(s+1), X(s+1) landed), lgkmcnt(0) (this wave's reads of buffer B are complete), barrier
    //   ks3: MFMA fb | read F(s+1,0) -> fa from buffer 1-B | decode + issue G(s+2) -> buffer B (free behind SYNC(s))
    auto step = [&](auto bufc, int s, auto has1, auto has2) {     // (has1 / has2: std::true_type in the steady-state loop, bool in the tail)
        constexpr int B = decltype(bufc)::value;
        using other_t = std::integral_constant<int, 1 - B>;
        WG_SUBSTEP(fa, fb, B, 1, true, if (has1) issue_x(other_t{}, c0_t{}), if (has1) issue_x(other_t{}, c2_t{}));
        WG_SUBSTEP(fb, fa, B, 2, true, (void)0, (void)0);
        WG_SUBSTEP(fa, fb, B, 3, true, (void)0, (void)0);
        if (has1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            WSB();
            __builtin_amdgcn_s_barrier();
            WSB();
        }
        WG_SUBSTEP(fb, fa, 1 - B, 0, has1, if (has2) issue_g(s + 2, bufc, c0_t{}), if (has2) issue_g(s + 2, bufc, c2_t{}));
    };
    // static priority for the younger half of the workgroup over the whole K loop (see conv_pipe.hip, round 3)
    if ((threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);
    {
        int s = 0;
        for (; s + 3 < nsteps; s += 2) {           // steady state: both K-steps of the pair have two successors
            step(c0_t{}, s, std::true_type{}, std::true_type{});
            step(c1_t{}, s + 1, std::true_type{}, std::true_type{});
        }
        for (; s < nsteps; s += 2) {
            step(c0_t{}, s, s + 1 < nsteps, s + 2 < nsteps);
            if (s + 1 < nsteps) step(c1_t{}, s + 1, s + 2 < nsteps, s + 3 < nsteps);
        }
    }
    __builtin_amdgcn_s_setprio(0);

    if (do_db && (lane & 31) == 0) {
        float* db = WGRAD_DB_PTR(a, grp);
        const float dbs = WGRAD_DB_SCALE(a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 128 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            unsafeAtomicAdd(db + n, accb[r] * dbs);
        }
    }
    const int sidx = PM ? (int)a.pm_split[bz] : bz;
    const int ns_t = PM ? a.pm_blk0[t + 1] - a.pm_blk0[t] : (int)gridDim.z;
    float* dw = a.dw + WGRAD_DW_OFF(a, grp) + (long)a.wt[t] * a.Cout * a.Cin + (a.partial ? (long)sidx * a.split_stride : 0l);
    const int nzero = (a.partial && sidx == ns_t - 1) ? a.nslots - ns_t : 0;      // this tap's unused slots
    const float osc = WGRAD_OUT_SCALE(a);          // (1 unless the operands are scaled planes: exact either way)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

// host side: launched from loft_conv_wgrad_bf16_v (conv_mfma.hip).  Requires Cout % 256 == 0 and Cin % 256 == 0.
int loft_launch_conv_wgrad_stream(const WgradArgs& a, dim3 grid, bool pm, hipStream_t s) {
    const bool samesize = a.gos == 1 && a.ss == 1 && a.GH == a.OH && a.GW == a.OW && a.XH == a.OH && a.XW == a.OW;
    bool same = samesize && a.OW >= 64;
    for (int t = 0; t < a.T; ++t) same = same && a.goy[t] == 0 && a.gox[t] == 0;
    const bool dense = samesize && a.T == 1 && a.goy[0] == 0 && a.gox[0] == 0 && a.dy[0] == 0 && a.dx[0] == 0;
    bool inc = pm && a.pm_inc_ok;
    for (int t = 0; t < a.T && inc; ++t) inc = a.pm_blk0[t + 1] == a.pm_blk0[t] || a.pm_pps[t] >= 64;
    if (pm && inc) hipLaunchKernelGGL(conv_wgrad_stream_kernel<1>, grid, dim3(512), 0, s, a);
    else if (pm) hipLaunchKernelGGL(conv_wgrad_stream_kernel<4>, grid, dim3(512), 0, s, a);
    else if (dense) hipLaunchKernelGGL(conv_wgrad_stream_kernel<2>, grid, dim3(512), 0, s, a);
    else if (same) hipLaunchKernelGGL(conv_wgrad_stream_kernel<3>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(conv_wgrad_stream_kernel<0>, grid, dim3(512), 0, s, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================
// conv_wgrad_ring_kernel -- the 128 x 128 tile (Cout / Cin multiples of 128 but not of 256, or too little K per tile for the
// 256 form: the backbone's 1x1 and 3x3 layers) with a FOUR-stage ring of 32-pixel K-steps instead of conv_wgrad_kernel<128,4>'s
// two 64-pixel stages.  The lock-step kernel prefetches one K-step ahead; one workgroup per CU then runs a K-step in ~2300
// cycles against 512 of MFMA issue (tools/probes/wgrad_splits.py) because an L2 / HBM round trip under load is longer than the
// one K-step of compute that is supposed to cover it.  Here three K-steps (96 pixels) are in flight behind the one being
// consumed, retired by a COUNTED s_waitcnt (vmcnt(8): this thread's four copies of the oldest of three stages), same 64 KiB of LDS, two
// workgroups per CU.  LDS: 4 x [G 32 px x 256 B][X 32 px x 256 B], 16-byte chunk q of pixel row r at q ^ ((r & 3) << 2).
// 2 x 2 waves of 64 (n) x 64 (c); per 16-pixel sub-step 2 G + 2 X fragments (transposing reads, inline asm), 4 MFMAs.
// =====================================================================================
namespace {
constexpr int RG_STAGE = 16384, RG_X = 8192, RG_LDS = 65536;
}

// DENSE: one tap at offset zero, unit strides, G / X / output maps of one size (every 1x1 layer of the backbone): reduction row
// m IS pixel m of both operands -- no per-row decode (two exact divisions, bounds tests and 64-bit address arithmetic per staged
// row: ~60 VALU instructions per 32-pixel K-step, more issue time than the step's 8 MFMAs at one wave per SIMD).
// SAME (MODE 2): unit strides, G / X / output maps of one size, taps with arbitrary offsets on the X side only (every 3x3 / pad 1
// layer): the G row of reduction row m is pixel m, the X row is pixel m + dy * W + dx when (oy + dy, ox + dx) is inside the map;
// (oy, ox) of the two rows a thread stages are carried from K-step to K-step (+32 pixels, at most one row wrap when W >= 32).
template <int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_ring_kernel(const WgradArgs a) {
    constexpr bool DENSE = MODE == 1, SAME = MODE == 2;
    __shared__ __attribute__((aligned(16))) char lds[RG_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int V = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk);
    const int bx = V % gridDim.x, by = (V / gridDim.x) % gridDim.y, bz = V / (gridDim.x * gridDim.y);
    const int nt = bx / a.ctiles, ct = bx - nt * a.ctiles;
    const int t = by % a.T, grp = by / a.T;
    const int mbeg = bz * a.pix_per_split, mend = min(a.M, mbeg + a.pix_per_split);
    const int n0 = nt * 128, c0 = ct * 128;
    if (mbeg >= mend) return;
    const bf16_t* G = a.g + WGRAD_G_OFF(a, grp);
    const bf16_t* X = a.x + WGRAD_X_OFF(a, grp);
    const int ohw = a.OH * a.OW;
    const int goy = a.goy[t], gox = a.gox[t], dy = a.dy[t], dx = a.dx[t];

    // staging: a wave-level copy = 4 pixel rows x 256 B; thread -> rows i*16 + wave*4 + (lane >> 4), i = 0, 1, chunk lane & 15
    const int lrow = lane >> 4, lchunk = lane & 15;
    int s_oy[2] = {0, 0}, s_ox[2] = {0, 0};           // SAME: map position of this thread's rows of the NEXT stage() call
    const int xshift = dy * a.XW + dx;
    if constexpr (SAME) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mbeg + i * 16 + wave * 4 + lrow;
            const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
            s_oy[i] = fastdiv(rem, a.ow_mul, a.ow_sh);
            s_ox[i] = rem - s_oy[i] * a.OW;
        }
    }
    auto stage = [&](int step, auto bufc) {
        constexpr int B = decltype(bufc)::value;
        char* gbuf = lds + B * RG_STAGE;
        char* xbuf = gbuf + RG_X;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 16 + wave * 4 + lrow;
            const int m = mbeg + step * 32 + row;
            const int q = wswz(row, lchunk) * 8;
            const bf16_t* pg = a.zero_page;
            const bf16_t* px = a.zero_page;
            if constexpr (DENSE) {
                if (m < mend) {
                    pg = G + (long)m * a.Cout + n0 + q;
                    px = X + (long)m * a.Cin + c0 + q;
                }
            } else if constexpr (SAME) {
                const int iy = s_oy[i] + dy, ix = s_ox[i] + dx;
                if ((m < mend) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                    pg = G + (long)m * a.Cout + n0 + q;
                    px = X + (long)(m + xshift) * a.Cin + c0 + q;
                }
                s_ox[i] += 32;                                  // the row this thread stages at the next K-step
                if (s_ox[i] >= a.OW) {
                    s_ox[i] -= a.OW;
                    s_oy[i] = s_oy[i] + 1 == a.OH ? 0 : s_oy[i] + 1;
                }
            } else
            if (m < mend) {
                const int b = fastdiv(m, a.ohw_mul, a.ohw_sh), rem = m - b * ohw;
                const int oy = fastdiv(rem, a.ow_mul, a.ow_sh), ox = rem - oy * a.OW;
                const int gy = oy * a.gos + goy, gx = ox * a.gos + gox;
                const int iy = oy * a.ss + dy, ix = ox * a.ss + dx;
                if ((gy >= 0) & (gy < a.GH) & (gx >= 0) & (gx < a.GW) & (iy >= 0) & (iy < a.XH) & (ix >= 0) & (ix < a.XW)) {
                    pg = G + ((long)(b * a.GH + gy) * a.GW + gx) * a.Cout + n0 + q;
                    px = X + ((long)(b * a.XH + iy) * a.XW + ix) * a.Cin + c0 + q;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)pg, (lds_ptr_t)(gbuf + (i * 16 + wave * 4) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)px, (lds_ptr_t)(xbuf + (i * 16 + wave * 4) * 256), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wn = wave >> 1, wc = wave & 1;
    const bool do_db = WGRAD_DB_ON(a, grp) && ct == 0 && (a.db_tap == -2 || a.db_tap == t);      // wave (wn, wc) takes n-block wn*2 + wc
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)LOFT_ONE16;

    // per-lane bases of the transposing reads (tr_frag_issue of conv_mfma.hip, rows of 256 B): rows r0 = 8*(gl>>1) + (il>>2) and
    // r0 + 4 of a 16-pixel sub-step, 16-byte chunk (col >> 3) ^ ((r & 3) << 2), + (col & 7) * 2
    const int il = lane & 15, gl = lane >> 4;
    const int r0 = 8 * (gl >> 1) + (il >> 2);
    unsigned gaddr[2], xaddr[2];           // fragment i of G / X: byte offset inside a stage's G / X tile for sub-step 0, row r0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int colg = wn * 64 + i * 32 + 16 * (gl & 1) + (il & 3) * 4;
        const int colx = wc * 64 + i * 32 + 16 * (gl & 1) + (il & 3) * 4;
        gaddr[i] = (unsigned)(size_t)(lds + r0 * 256 + wswz(r0, colg >> 3) * 16 + (colg & 7) * 2);
        xaddr[i] = (unsigned)(size_t)(lds + RG_X + r0 * 256 + wswz(r0, colx >> 3) * 16 + (colx & 7) * 2);
    }
    // (rows r0 + 4, r0 + 16, r0 + 20 have the same (row & 3) as r0: their chunks differ only through the row offset)

    const int nsteps = (mend - mbeg + 31) / 32;
    using b0_t = std::integral_constant<int, 0>;
    using b1_t = std::integral_constant<int, 1>;
    using b2_t = std::integral_constant<int, 2>;
    using b3_t = std::integral_constant<int, 3>;
    stage(0, b0_t{}); stage(1, b1_t{}); stage(2, b2_t{});           // (steps past the range copy the zero page: uniform counts)
    // Fragments are read ONE SUB-STEP AHEAD of their MFMAs (two register sets), also across the K-step boundary: with one wave per
    // SIMD and workgroup the read -> wait -> MFMA chain of a sub-step was ~400 cycles for 128 of MFMA issue.
    s16x4 lo[2][4], hi[2][4];
    // (a macro, not a generic lambda: clang rejects inline-asm operands that name captured locals inside one)
#define RG_RD(S, B, KS)                                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                             \
        TR_READ(lo[S][i_], gaddr[i_], (B) * RG_STAGE + (KS) * 4096);                                                               \
        TR_READ(hi[S][i_], gaddr[i_], (B) * RG_STAGE + (KS) * 4096 + 1024);                                                        \
        TR_READ(lo[S][2 + i_], xaddr[i_], (B) * RG_STAGE + (KS) * 4096);                                                           \
        TR_READ(hi[S][2 + i_], xaddr[i_], (B) * RG_STAGE + (KS) * 4096 + 1024);                                                    \
    }
    auto mm = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        bf16x8 f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i][0] = lo[S][i][0]; f[i][1] = lo[S][i][1]; f[i][2] = lo[S][i][2]; f[i][3] = lo[S][i][3];
            f[i][4] = hi[S][i][0]; f[i][5] = hi[S][i][1]; f[i][6] = hi[S][i][2]; f[i][7] = hi[S][i][3];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = LOFT_MFMA_32x32x16(f[i], f[2 + j], acc[i][j]);
        if (do_db) accb = LOFT_MFMA_32x32x16(wc == 0 ? f[0] : f[1], ones, accb);
    };
#define RG_WAIT(S) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[S][0]), "+v"(lo[S][1]), "+v"(lo[S][2]), "+v"(lo[S][3]), \
                                "+v"(hi[S][0]), "+v"(hi[S][1]), "+v"(hi[S][2]), "+v"(hi[S][3]))
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                // stage 0 landed (this thread's copies) ...
    __syncthreads();                                                // ... and everybody's
    RG_RD(0, 0, 0);                                                 // F(0, 0) -> set 0
    // K-step s (stage buffer B, stage s visible, set 0 = F(s, 0) in flight):
    //   read F(s,1) -> set 1 | MFMA set 0 | own copies of stage s+1 landed (vmcnt 4), own reads of stage s complete | barrier
    //   | issue stage s+3 (into the buffer of stage s-1) | read F(s+1,0) -> set 0 | MFMA set 1
#define RG_STEP(s, B, NB, FREEC)                                                                     \
    do {                                                                                             \
        RG_WAIT(0);                                                                                  \
        RG_RD(1, B, 1);                                                                              \
        WSB();                                                                                       \
        mm(b0_t{});                                                                                  \
        WSB();                                                                                       \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
        RG_WAIT(1);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                \
        WSB();                                                                                       \
        stage((s) + 3, FREEC);                                                                       \
        RG_RD(0, NB, 0);                                                                             \
        WSB();                                                                                       \
        mm(b1_t{});                                                                                  \
        WSB();                                                                                       \
    } while (0)
    for (int s = 0; s < nsteps; s += 4) {
        RG_STEP(s, 0, 1, b3_t{});
        if (s + 1 < nsteps) RG_STEP(s + 1, 1, 2, b0_t{});
        if (s + 2 < nsteps) RG_STEP(s + 2, 2, 3, b1_t{});
        if (s + 3 < nsteps) RG_STEP(s + 3, 3, 0, b2_t{});
    }
    RG_WAIT(0);
#undef RG_WAIT
#undef RG_STEP
#undef RG_RD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (zero-page copies of the steps past the range)

    if (do_db && (lane & 31) == 0) {
        float* db = WGRAD_DB_PTR(a, grp);
        const float dbs = WGRAD_DB_SCALE(a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 64 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            unsafeAtomicAdd(db + n, accb[r] * dbs);
        }
    }
    float* dw = a.dw + WGRAD_DW_OFF(a, grp) + (long)a.wt[t] * a.Cout * a.Cin + (a.partial ? (long)bz * a.split_stride : 0l);
    const float osc = WGRAD_OUT_SCALE(a);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float* p = dw + (long)n * a.Cin + c;
                if (a.partial) *p = acc[i][j][r];
                else unsafeAtomicAdd(p, acc[i][j][r] * osc);
            }
        }
}

int loft_launch_conv_wgrad_ring(const WgradArgs& a, dim3 grid, hipStream_t s) {
    const bool dense = a.T == 1 && a.gos == 1 && a.ss == 1 && a.goy[0] == 0 && a.gox[0] == 0 && a.dy[0] == 0 && a.dx[0] == 0 &&
                       a.GH == a.OH && a.GW == a.OW && a.XH == a.OH && a.XW == a.OW;
    bool same = a.gos == 1 && a.ss == 1 && a.GH == a.OH && a.GW == a.OW && a.XH == a.OH && a.XW == a.OW && a.OW >= 32;
    for (int t = 0; t < a.T; ++t) same = same && a.goy[t] == 0 && a.gox[t] == 0;
    if (dense) hipLaunchKernelGGL(conv_wgrad_ring_kernel<1>, grid, dim3(256), 0, s, a);
    else if (same) hipLaunchKernelGGL(conv_wgrad_ring_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv_wgrad_ring_kernel<0>, grid, dim3(256), 0, s, a);
    LOFT_LAUNCH_CHECK();
    return 0;
}
