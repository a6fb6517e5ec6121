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
                                               (8, 512, 128, 128, 1, 1),      # deep 1x1
                                               (3524, 256, 7, 7, 3, 1),       # FOA head maps (881 positives x 4 branches): pixel-major
                                               (881, 256, 14, 14, 3, 1)])     # mask head maps    tiles, valid-rows-only weight gradient
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


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def test_resample_adjoints_fullsize():
    """HRFPN / HRModule resampling kernels at BASELINE config-5 sizes (8 x 1024^2 input -> 256^2 maps): each backward is
    the exact adjoint of its forward, <F(x), g> == <x, F^T(g)> (fp32, relative 1e-4), and the fuse kernel is linear where its
    ReLU is inactive."""
    from bonai_amd import kernels as K
    g = torch.Generator(device='cuda').manual_seed(0)
    B, H = 8, 256

    def rnd(*shape):
        return torch.randn(*shape, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    # bilinear x4 into a slot of a 512-channel tensor
    x = rnd(B, 128, H // 4, H // 4)
    gout = rnd(B, 512, H, H)
    dst = torch.zeros_like(gout)
    K.bilinear_up_slot_(x, dst, 2, 128)
    gx = K.bilinear_up_slot_bwd(gout, 128, 2, 128)
    lhs, rhs = _dot(dst, gout), _dot(x, gx)
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    assert dst[:, :128].abs().max().item() == 0 and dst[:, 256:].abs().max().item() == 0      # only its slot is written
    # avg-pool x8
    x = rnd(B, 256, H, H)
    y = K.avgpool(x, 3)
    gy = torch.randn_like(y)
    lhs, rhs = _dot(y, gy), _dot(x, K.avgpool_bwd(gy, 3))
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    # fuse: sum of a fine and a x4-coarser term, shifted so the ReLU never clips -> linear; backward = adjoint
    a, b = rnd(B, 64, H, H).abs() + 5.0, rnd(B, 64, H // 4, H // 4).abs() + 5.0
    y = K.fuse_sum_relu([a, b], [0, 2])
    gy = torch.randn_like(y)
    lhs = _dot(y, gy)
    rhs = _dot(a, K.blocksum_masked(gy, y, 0)) + _dot(b, K.blocksum_masked(gy, y, 2))
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


def test_dcn_sampling_adjoint_fullsize():
    """DCNv2 sampling at the config-4 layer3 shape (8 x 256 x 64^2, 3x3), bf16 binned backward: for fixed offsets the map
    x -> col is linear, so <col(x), g> == <x, dx(g)> up to bf16 rounding of the operands (1e-2 relative); the offset
    gradient matches a central finite difference of <col, g> along a random direction."""
    from bonai_amd import kernels as K
    g = torch.Generator(device='cuda').manual_seed(1)
    B, C, H = 8, 256, 64
    x = torch.randn(B, C, H, H, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    om = torch.zeros(B, 28, H, H, device='cuda').contiguous(memory_format=torch.channels_last)
    om[:, :27] = torch.randn(B, 27, H, H, device='cuda', generator=g) * 0.7
    # offsets = integer + fraction in [0.25, 0.75]: +-eps never crosses a pixel boundary, where the bilinear sample has a kink
    om[:, :18] = torch.randint(-1, 2, (B, 18, H, H), device='cuda', generator=g).float() + \
        0.25 + 0.5 * torch.rand(B, 18, H, H, device='cuda', generator=g)
    col = K.mdcn_sample_fwd(x, om, 3, 3, 1, 1)
    gcol = torch.randn(col.shape, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dx, dom = K.mdcn_sample_bwd(x, om, gcol, 3, 3, 1, 1)
    lhs, rhs = _dot(col, gcol), _dot(x, dx)
    assert abs(lhs - rhs) <= 1e-2 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    # directional derivative w.r.t. the raw conv_offset output (fp32 sampling of the same bf16-valued x for a clean difference)
    xf, gf = x.float(), gcol.float()
    d = torch.zeros_like(om)
    d[:, :27] = torch.rand(B, 27, H, H, device='cuda', generator=g) * 2 - 1
    eps = 1e-2
    fp = _dot(K.mdcn_sample_fwd(xf, om + eps * d, 3, 3, 1, 1), gf)
    fm = _dot(K.mdcn_sample_fwd(xf, om - eps * d, 3, 3, 1, 1), gf)
    fd = (fp - fm) / (2 * eps)
    an = _dot(dom, d)
    assert abs(fd - an) <= 0.03 * max(abs(fd), abs(an), 1.0), (fd, an)
