"""A/B timing of the pipelined 256x256 tap-conv variants (conv_pipe.hip VAR bits) and a barrier-level time line.

    python tools/pipe_trace.py            # run on the GPU box

VAR bits ride in bits 12-15 of the C-ABI variant argument: 1 TRACE, 2 NOPRIO, 4 OLDORDER.  TRACE: lane 0 of every wave of
workgroups 0 and 1 stores s_memtime at every barrier exit of K-tiles 8..11 (the buffer is passed in place of the bias)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bonai_amd import kernels as K

SHAPES = [('fpn.P2.3x3', 8, 256, 256, 256, 256, 3, 1, 1, 1), ('mask.3x3', 872, 256, 256, 14, 14, 3, 1, 1, 1),
          ('foa.3x3', 3488, 256, 256, 7, 7, 3, 1, 1, 4), ('layer3.3x3', 8, 256, 256, 64, 64, 3, 1, 1, 1),
          ('fc1', 8192, 12544, 1024, 1, 1, 1, 1, 0, 1)]


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    if os.environ.get('R6'):          # round 6: the activations-first schedule and the lean instruction stream against the shipped kernel
        variants = [('stream0', K.CONV_STREAM256), ('xfirst', K.CONV_XFIRST), ('lean', K.CONV_LEAN), ('leanx', K.CONV_LEANX),
                    ('stream8', K.CONV_STREAM256 | (8 << 12))]
    else:
      variants = ([('ring32', K.CONV_RING32), ('w4', K.CONV_W4), ('stream_direct', K.CONV_STREAM256 | (2 << 12) | K.CONV_FLAG_NO_ROI_BLOCKS)] if os.environ.get('RING32') else [('lockstep', K.CONV_T256_FAST), ('stream_tapmajor', K.CONV_STREAM256 | K.CONV_FLAG_TAP_MAJOR),
                ('stream_krot', K.CONV_STREAM256 | K.CONV_FLAG_KROT)]) + [(f'stream{v}', K.CONV_STREAM256 | (v << 12)) for v in
                                                   [int(x) for x in (os.environ.get('STREAM_VARS') or '0').split(',')]] + [(f'pipe{v}', K.CONV_PIPE256 | (v << 12)) for v in
                                                   [int(x) for x in (os.environ.get('PIPE_VARS') or '0,2,4').split(',')]]
    print(f'{"shape":14s}' + ''.join(f'{n + " TF":>14s}' for n, _ in variants))
    for name, B, Cin, Cout, H, W, R, st, pad, G in SHAPES:
        x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(G, Cout, Cin, R, R, device='cuda') * 0.02
        wp = torch.stack([K.pack_w_fwd(w[i]) for i in range(G)])
        bias = torch.zeros(G, Cout, device='cuda')
        gflop = 2.0 * B * H * W * Cout * Cin * R * R / 1e9
        best = {n: 1e9 for n, _ in variants}
        for rnd in range(3):                       # interleaved rounds, best of three
            for n, v in variants:
                K.CONV_VARIANT = v
                best[n] = min(best[n], timeit(lambda: K.conv2d_fwd(x, wp, bias, R, R, st, pad, relu=True, groups=G)))
                K.CONV_VARIANT = K.CONV_AUTO
        print(f'{name:14s}' + ''.join(f'{gflop / best[n]:14.1f}' for n, _ in variants), flush=True)
    if os.environ.get('STREAM_TRACE'):
        name, B, Cin, Cout, H, W, R, st, pad, G = SHAPES[0]
        if os.environ.get('TRACE_SHAPE'):       # e.g. TRACE_SHAPE=8,256,1024,64,64,1,1,0,1 (B,Cin,Cout,H,W,R,stride,pad,groups)
            B, Cin, Cout, H, W, R, st, pad, G = [int(v) for v in os.environ['TRACE_SHAPE'].split(',')]
        x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        wp = K.pack_w_fwd(torch.randn(Cout, Cin, R, R, device='cuda') * 0.02)[None]
        tr = torch.zeros(2 * 8 * 64, dtype=torch.int64, device='cuda')
        K.CONV_VARIANT = K.CONV_STREAM256 | (int(os.environ.get('STREAM_TRACE_VAR', '1')) << 12)
        K.conv2d_fwd(x, wp, tr.view(torch.float32), R, R, st, pad, groups=G)
        K.CONV_VARIANT = K.CONV_AUTO
        torch.cuda.synchronize()
        raw = tr.cpu().numpy().reshape(2, 8, 64)[0]
        print('stream trace, K-tiles 8 and 9; per tile: ks0 [start, after 1st MFMA pair], ks1 [..], ks2 [..], sync [before wait, '
              'after wait], ks3 [after barrier, after 1st pair], end')
        have_tiles = bool((raw[:, :24] > 0).any())      # (K-shallow launches have no K-tiles 8 / 9: kernel-level stamps only)
        t0 = raw[:, :24][raw[:, :24] > 0].min() if have_tiles else 0
        for wv in range(8):
            r = raw[wv, :24].astype(np.int64)
            for tl in range(2 if have_tiles else 0):
                v = r[12 * tl:12 * tl + 11] - t0
                print(f'w{wv} simd{(int(raw[wv, 63]) >> 4) & 3} tile{8 + tl}: ' + ' '.join(f'{int(q):6d}' for q in v))
            k = raw[wv, 32:37].astype(np.int64)
            print(f'w{wv} kernel: setup {k[4] - k[0]}  first loads {k[1] - k[4]}  loop {k[2] - k[1]} ({(k[2] - k[1]) / max(1, R * R * Cin // 64):.0f} per K-tile)  epilogue+drain {k[3] - k[2]}')
            e = raw[wv, 40:49].astype(np.int64)
            print(f'   setup: entry->decode {e[0] - k[0]}  decode {e[1] - e[0]}  masks {e[2] - e[1]}  rest {k[4] - e[2]} | epilogue: '
                  f'sync {e[4] - k[2]}  bias+relu {e[5] - e[4]}  lds write {e[6] - e[5]}  barrier {e[7] - e[6]}  stores issued {e[8] - e[7]}  drain {k[3] - e[8]}')
        ms = timeit(lambda: K.conv2d_fwd(x, wp, None, R, R, st, pad, groups=G))
        print(f'P2 3x3 launch {ms * 1e3:.1f} us = 8 rounds of 36 K-tiles per CU')
    if os.environ.get('PIPE_NOTRACE'):
        return
    # ---- time line of the P2 3x3
    name, B, Cin, Cout, H, W, R, st, pad, G = SHAPES[0]
    x = torch.randn(B, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    wp = K.pack_w_fwd(torch.randn(Cout, Cin, R, R, device='cuda') * 0.02)[None]
    for var in (1, 5):
        tr = torch.zeros(2 * 8 * 64, dtype=torch.int64, device='cuda')
        K.CONV_VARIANT = K.CONV_PIPE256 | (var << 12)
        K.conv2d_fwd(x, wp, tr.view(torch.float32), R, R, st, pad, groups=G)
        K.CONV_VARIANT = K.CONV_AUTO
        torch.cuda.synchronize()
        raw = tr.cpu().numpy().reshape(2, 8, 64)[0]
        hw = raw[:, 63]
        print(f'--- VAR={var}: HW_ID simd per wave: ' + ' '.join(f'w{w}:simd{(int(h) >> 4) & 3}/cu{(int(h) >> 8) & 15}' for w, h in enumerate(hw)))
        t = raw[:, :60].astype(np.int64)                     # 3 K-tiles x 4 phases x [L_end, M_exit, M_start, M_end, L_begin(next)]
        t0 = t.min()
        ev = (t - t0).reshape(8, 12, 5)
        names = ['L', 'bar>M', 'lgkm', 'mfma', 'bar>L']
        print('per phase (cycles, mean over the 4 waves of a group): L = reads+copies issued, bar>M = wait at the barrier into the'
              ' MFMA half, lgkm = fragment wait, mfma = 8 MFMAs issued, bar>L = wait at the barrier out of it')
        for grp in (0, 1):
            e = ev[4 * grp:4 * grp + 4].mean(0)             # [12 phases][5]
            prev_lb = None
            for ph in range(12):
                L = e[ph, 0] - prev_lb if prev_lb is not None else float('nan')
                row = [L, e[ph, 1] - e[ph, 0], e[ph, 2] - e[ph, 1], e[ph, 3] - e[ph, 2], e[ph, 4] - e[ph, 3]]
                prev_lb = e[ph, 4]
                print(f'g{grp} tile{8 + ph // 4} p{ph % 4}: ' + ' '.join(f'{n}={v:5.0f}' for n, v in zip(names, row)) +
                      f'   | M_exit at {e[ph, 1]:6.0f}')
        print(f'cycles per K-tile: {(ev[:, 8:, 1].mean() - ev[:, :4, 1].mean()) / 2:.0f}  (MFMA-bound minimum 2048)')


if __name__ == '__main__':
    main()
