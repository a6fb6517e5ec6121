"""GPU, BASELINE full sizes (batch 8, 1024x1024 pyramid): size-independent properties instead of an (hours-long)
CPU oracle run -- the three conv kernels must be mutually adjoint,
        <conv(x, w), g>  ==  <x, dgrad(g, w)>  ==  <w, wgrad(g, x)>,
linear in their data operand, and NMS at the full 12 768-candidate size must be idempotent and score-ordered."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('B,C,H,W,R,stride', [(8, 256, 256, 256, 3, 1),      # FPN P2 / RPN conv: M = 524 288
                                               (8, 256, 128, 128, 3, 2),      # stride-2 3x3 (layer3.0.conv2 shape)
                                               (8, 512, 128, 128, 1, 1)])     # deep 1x1
def test_conv_triple_adjoint_full_size(B, C, H, W, R, stride):
    from bonai_amd import kernels as K
    torch.manual_seed(0)
    Cout = 256
    pad = R // 2
    x = _cl(torch.randn(B, C, H, W, device='cuda').bfloat16())
    w = (torch.randn(Cout, C, R, R, device='cuda') / (C * R * R) ** 0.5).bfloat16().float()
    y = K.conv2d_fwd(x, K.pack_w_fwd(w)[None], None, R, R, stride, pad, out_dtype=torch.float32)
    g = _cl(torch.randn_like(y).bfloat16())
    gx = K.conv2d_dgrad(g, K.pack_w_dgrad(w)[None], (H, W), R, R, stride, pad, out_dtype=torch.float32)
    dw = K.unpack_dw(K.conv2d_wgrad(g, x, R, R, stride, pad)[0], w.shape)
    a = (y.double() * g.double()).sum().item()
    b = (x.double() * gx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    scale = (y.double().pow(2).sum().sqrt() * g.double().pow(2).sum().sqrt()).item()
    assert abs(a - b) < 2e-4 * scale and abs(a - c) < 2e-4 * scale, (a, b, c, scale)
    # linearity in x (fp32 accumulation of bf16 products is exact up to summation order)
    x2 = _cl(torch.randn_like(x))
    y12 = K.conv2d_fwd(_cl((x.float() + x2.float()).bfloat16()), K.pack_w_fwd(w)[None], None, R, R, stride, pad,
                       out_dtype=torch.float32)
    y2 = K.conv2d_fwd(x2, K.pack_w_fwd(w)[None], None, R, R, stride, pad, out_dtype=torch.float32)
    # (x + x2) is re-rounded to bf16: compare against the conv of exactly that rounded sum via linearity of the residual
    resid = _cl(((x.float() + x2.float()).bfloat16().float() - x.float() - x2.float()).bfloat16())
    yr = K.conv2d_fwd(resid, K.pack_w_fwd(w)[None], None, R, R, stride, pad, out_dtype=torch.float32)
    err = (y12 - (y + y2 + yr)).abs().max().item()
    assert err < 5e-3 * max(1.0, y12.abs().max().item()), err


def test_nms_full_size_properties():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(11)
    sizes = [3000, 3000, 3000, 3000, 768] * 8            # 8 images x 5 levels = BASELINE batch
    n = sum(sizes)
    c = rng.uniform(0, 1024, (n, 2)); wh = rng.uniform(8, 256, (n, 2))
    boxes = torch.tensor(np.concatenate([c - wh / 2, c + wh / 2], 1).clip(0, 1024), dtype=torch.float32).cuda()
    scores = torch.tensor(rng.rand(n), dtype=torch.float32).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64).cuda()
    ks, order = K.segmented_sort_desc(scores, off)
    order = order.long()
    assert all(bool((ks[a:b][:-1] >= ks[a:b][1:]).all()) for a, b in zip(off[:-1].tolist(), off[1:].tolist()))
    sb = boxes[order]
    keep = K.nms_segmented(sb, off, 0.7).bool()
    # idempotence: NMS of the survivors (same segments) keeps everything
    counts = torch.stack([keep[a:b].sum() for a, b in zip(off[:-1].tolist(), off[1:].tolist())])
    off2 = torch.cat([torch.zeros(1, dtype=torch.int64, device='cuda'), counts.cumsum(0)])
    keep2 = K.nms_segmented(sb[keep], off2, 0.7).bool()
    assert bool(keep2.all())
    # and the first box of every non-empty segment always survives
    for a, b in zip(off[:-1].tolist(), off[1:].tolist()):
        assert b == a or bool(keep[a])
