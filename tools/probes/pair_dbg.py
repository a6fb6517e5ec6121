import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bonai_amd import kernels as K
def cl(t): return t.to('cuda', torch.bfloat16).contiguous(memory_format=torch.channels_last)
def rows(t): return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).double()
for (B, P, H, W) in [(1, 256, 16, 16), (2, 256, 8, 24), (2, 128, 16, 32)]:
    torch.manual_seed(5)
    C = 4 * P
    t2 = cl(torch.randn(B, P, H, W).relu()); x = cl(torch.randn(B, C, H, W).relu())
    w3 = torch.randn(C, P, 1, 1, device='cuda') / P ** 0.5
    w1n = torch.randn(P, C, 1, 1, device='cuda') / C ** 0.5
    b3, b1n = torch.randn(C, device='cuda') * 0.1, torch.randn(P, device='cuda') * 0.1
    wp3, wp1n = K.pack_w_fwd(w3), K.pack_w_fwd(w1n)
    out_s = K.conv2d_fwd(t2, wp3[None], b3[None], 1, 1, 1, 0, relu=True, residual=x)
    (k3, k1), _ = K.pack_k8([wp3, wp1n])
    out_f, t1_f = K.bneck_pair(t2, k3, b3, x, k1, b1n)
    t1_s = K.conv2d_fwd(out_f, wp1n[None], b1n[None], 1, 1, 1, 0, relu=True)
    ref = (rows(t2) @ wp3[0].double().t() + b3.double() + rows(x)).relu()
    ref2 = (rows(out_f) @ wp1n[0].double().t() + b1n.double()).relu()
    for nm, f, s, r in (('mid', out_f, out_s, ref), ('out2', t1_f, t1_s, ref2)):
        f_, s_ = rows(f), rows(s)
        d = (f_ - s_).abs()
        ulp = torch.maximum(f_.abs(), s_.abs()) * 2.0 ** -7 + 1e-30
        bad = d > 1.01 * ulp
        print(f'{(B,P,H,W)} {nm}: differing {int((d > 0).sum())} of {d.numel()}, > 1 ulp: {int(bad.sum())}; '
              f'max |fused - ref64| {float((f_ - r).abs().max()):.4g}, max |separate - ref64| {float((s_ - r).abs().max()):.4g}')
        idx = bad.nonzero()[:6]
        for (i, j) in idx.tolist():
            print(f'    row {i} ch {j}: fused {float(f_[i, j]):.6f} separate {float(s_[i, j]):.6f} ref64 {float(r[i, j]):.6f}')

    # backward form
    g_t1 = cl(torch.randn(B, P, H, W)); g_sc = cl(torch.randn(B, C, H, W)); out_k = cl(torch.randn(B, C, H, W).relu())
    wpt1n, wpt3 = K.pack_w_dgrad(w1n), K.pack_w_dgrad(w3)
    gx_s = K.conv2d_dgrad(g_t1, wpt1n[None], (H, W), 1, 1, 1, 0, residual=g_sc, mask=out_k)
    gt2_s = K.conv2d_dgrad(gx_s, wpt3[None], (H, W), 1, 1, 1, 0, mask=t2)
    (kk1, kk3), _ = K.pack_k8([wpt1n, wpt3])
    gx_f, gt2_f = K.bneck_pair(g_t1, kk1, None, g_sc, kk3, None, mask1=out_k, mask2=t2)
    print(f'{(B,P,H,W)} backward: mid equal {torch.equal(gx_f, gx_s)}  out2 equal {torch.equal(gt2_f, gt2_s)}; forward: mid equal {torch.equal(out_f, out_s)}')
    for nm, f, s_ in (('bwd mid', gx_f, gx_s), ('bwd out2', gt2_f, gt2_s), ('bwd out2 on fused mid', gt2_f, K.conv2d_dgrad(gx_f, wpt3[None], (H, W), 1, 1, 1, 0, mask=t2))):
        a_, b_ = rows(f), rows(s_)
        d = (a_ - b_).abs()
        nz = d > 0
        print(f'    {nm}: differing {int(nz.sum())} of {d.numel()}; max diff {float(d.max()):.5f}; max rel {float((d / (b_.abs() + 1e-9))[nz].max()) if nz.any() else 0:.5f}')
    ref_gx = (rows(g_t1) @ wpt1n[0].double().t() + rows(g_sc)) * (rows(out_k) > 0)
    print(f'    bwd mid vs ref64: fused {float((rows(gx_f) - ref_gx).abs().max()):.5f} separate {float((rows(gx_s) - ref_gx).abs().max()):.5f}; '
          f'rms fused {float((rows(gx_f) - ref_gx).pow(2).mean().sqrt()):.6f} separate {float((rows(gx_s) - ref_gx).pow(2).mean().sqrt()):.6f}')
