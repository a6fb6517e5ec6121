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
    out_f, t1_f = K.bneck_pair(t2, wp3, b3, x, wp1n, b1n)
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
