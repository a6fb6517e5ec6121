"""loft_bneck_pair_bf16 against the two launches it replaces, at the bench's layer2 / layer3 sizes (8 x 1024^2 input): microseconds per
pair, forward and backward, and the HBM floor of the fused form (operands once at 6 TB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def cl(t):
    return t.to('cuda', torch.bfloat16).contiguous(memory_format=torch.channels_last)


for name, B, P, H in (('layer3', 8, 256, 64), ('layer2', 8, 128, 128)):
    C = 4 * P
    t2, x = cl(torch.randn(B, P, H, H).relu()), cl(torch.randn(B, C, H, H).relu())
    w3 = K.pack_w_fwd(torch.randn(C, P, 1, 1, device='cuda') / P ** 0.5)
    w1 = K.pack_w_fwd(torch.randn(P, C, 1, 1, device='cuda') / C ** 0.5)
    b3, b1 = torch.randn(C, device='cuda'), torch.randn(P, device='cuda')
    hold = {}
    (k3, k1), _ = K.pack_k8([w3, w1])

    def sep_f():
        o = K.conv2d_fwd(t2, w3[None], b3[None], 1, 1, 1, 0, relu=True, residual=x)
        hold['o'] = K.conv2d_fwd(o, w1[None], b1[None], 1, 1, 1, 0, relu=True)

    def fus_f():
        hold['o'] = K.bneck_pair(t2, k3, b3, x, k1, b1)

    g1, gs = cl(torch.randn(B, P, H, H)), cl(torch.randn(B, C, H, H))
    wt1 = K.pack_w_dgrad(torch.randn(P, C, 1, 1, device='cuda') / C ** 0.5)
    wt3 = K.pack_w_dgrad(torch.randn(C, P, 1, 1, device='cuda') / P ** 0.5)

    (kt1, kt3), _ = K.pack_k8([wt1, wt3])

    def sep_b():
        gx = K.conv2d_dgrad(g1, wt1[None], (H, H), 1, 1, 1, 0, residual=gs, mask=x)
        hold['o'] = K.conv2d_dgrad(gx, wt3[None], (H, H), 1, 1, 1, 0, mask=t2)

    def fus_b():
        hold['o'] = K.bneck_pair(g1, kt1, None, gs, kt3, None, mask1=x, mask2=t2)

    M = B * H * H
    mb_f = M * (2 * P * 2 + 2 * C * 2) / 1e6
    mb_b = M * (3 * P * 2 + 3 * C * 2) / 1e6
    gf = 2.0 * M * 2 * P * C / 1e9
    for tag, sep, fus, mb in (('fwd', sep_f, fus_f, mb_f), ('bwd', sep_b, fus_b, mb_b)):
        a, b = timeit(sep), timeit(fus)
        a2, b2 = timeit(sep), timeit(fus)
        a, b = min(a, a2), min(b, b2)
        print(f'{name} {tag}: separate {a:7.1f} us   fused {b:7.1f} us ({gf / b * 1e3:6.1f} TFLOP/s, {mb / b:5.2f} TB/s of {mb:.0f} MB; floor at 6 TB/s {mb / 6.0:5.1f} us)')

    if P == 256:
        for v, what in ((1, 'no product 1'), (2, 'no product 2'), (3, 'no MFMA at all'), (4, 'no store of mid'), (8, 'no residual re-load'),
                        (12, 'no store, no residual'), (16, 'no epilogue 1'), (19, 'no MFMA, no epilogue 1'), (32, 'weights loaded once')):
            t = min(timeit(lambda: K.bneck_pair(t2, k3, b3, x, k1, b1, variant=v)) for _ in range(2))
            print(f'   ablation {v:2d} ({what}): {t:7.1f} us')
