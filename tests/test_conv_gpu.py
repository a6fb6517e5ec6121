"""GPU parity: the MFMA tap-convolution family vs torch-CPU fp32 on the same bf16-rounded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(t):  # bf16 rounding, kept in fp32 for the reference
    return t.bfloat16().float()


def _cl(t, dtype=torch.bfloat16):
    return t.to('cuda', dtype).contiguous(memory_format=torch.channels_last)


CASES = [
    # B, Cin, Cout, H, W, R, stride, pad
    (2, 64, 64, 20, 24, 1, 1, 0),
    (2, 64, 128, 20, 24, 3, 1, 1),
    (1, 128, 256, 17, 19, 3, 1, 1),     # ragged M
    (2, 256, 128, 16, 16, 3, 2, 1),
    (2, 256, 512, 16, 16, 1, 2, 0),
    (3, 256, 256, 7, 7, 3, 1, 1),       # RoI-sized maps
    (1, 128, 16, 12, 12, 1, 1, 0),      # narrow N (RPN cls/reg style, padded to 16)
]


@pytest.mark.parametrize('B,Cin,Cout,H,W,R,stride,pad', CASES)
def test_conv_fwd(B, Cin, Cout, H, W, R, stride, pad):
    from bonai_amd import kernels as K
    torch.manual_seed(1)
    x = _r(torch.randn(B, Cin, H, W))
    w = _r(torch.randn(Cout, Cin, R, R) / (Cin * R * R) ** 0.5)
    b = torch.randn(Cout)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    res = _r(torch.randn_like(ref))
    out = K.conv2d_fwd(_cl(x), K.pack_w_fwd(w.cuda()), b.cuda(), R, R, stride, pad, out_dtype=torch.float32)
    assert out.shape == ref.shape
    assert (out.cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    out2 = K.conv2d_fwd(_cl(x), K.pack_w_fwd(w.cuda()), b.cuda(), R, R, stride, pad, relu=True, residual=_cl(res))
    ref2 = F.relu(ref + res)
    assert (out2.float().cpu() - ref2).abs().max().item() < 1e-2 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize('B,Cin,Cout,H,W,R,stride,pad', [c for c in CASES if c[2] % 64 == 0])
def test_conv_dgrad(B, Cin, Cout, H, W, R, stride, pad):
    from bonai_amd import kernels as K
    torch.manual_seed(2)
    x = torch.randn(B, Cin, H, W, requires_grad=True)
    w = _r(torch.randn(Cout, Cin, R, R) / (Cin * R * R) ** 0.5)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    g = _r(torch.randn_like(y))
    (ref,) = torch.autograd.grad(y, x, g)
    out = K.conv2d_dgrad(_cl(g), K.pack_w_dgrad(w.cuda()), (H, W), R, R, stride, pad, out_dtype=torch.float32)
    assert (out.cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('B,Cin,Cout,H,W,R,stride,pad', [c for c in CASES if c[1] % 128 == 0 and c[2] % 128 == 0])
def test_conv_wgrad(B, Cin, Cout, H, W, R, stride, pad):
    from bonai_amd import kernels as K
    torch.manual_seed(3)
    x = _r(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, R, R, requires_grad=True)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    g = _r(torch.randn_like(y))
    (ref,) = torch.autograd.grad(y, w, g)
    for splits in (0, 1, 3):
        dwp = K.conv2d_wgrad(_cl(g), _cl(x), R, R, stride, pad, splits=splits)
        got = K.unpack_dw(dwp[0], w.shape).cpu()
        assert (got - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item()), splits
    dwp, db = K.conv2d_wgrad(_cl(g), _cl(x), R, R, stride, pad, with_bias=True)   # fused bias gradient
    want_b = g.sum(dim=(0, 2, 3))
    assert (db[0].cpu() - want_b).abs().max().item() < 1e-3 * max(1.0, want_b.abs().max().item())
    assert (K.unpack_dw(dwp[0], w.shape).cpu() - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())


def test_conv_grouped_matches_loop():
    """FOA-style: 4 independent branches in one launch (blockIdx.z)."""
    from bonai_amd import kernels as K
    torch.manual_seed(4)
    G, B, C = 4, 5, 256
    x = _r(torch.randn(G * B, C, 7, 7))
    w = _r(torch.randn(G, C, C, 3, 3) / (C * 9) ** 0.5)
    b = torch.randn(G, C)
    wp = torch.stack([K.pack_w_fwd(w[i].cuda()) for i in range(G)])
    out = K.conv2d_fwd(_cl(x), wp, b.cuda().contiguous(), 3, 3, 1, 1, relu=True, out_dtype=torch.float32, groups=G)
    for i in range(G):
        ref = F.relu(F.conv2d(x[i * B:(i + 1) * B], w[i], b[i], padding=1))
        assert (out[i * B:(i + 1) * B].cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_linear_as_conv():
    from bonai_amd import kernels as K
    torch.manual_seed(5)
    N, Kd, O = 300, 12544, 1024
    x = _r(torch.randn(N, Kd))
    w = _r(torch.randn(O, Kd) / Kd ** 0.5)
    b = torch.randn(O)
    ref = F.relu(F.linear(x, w, b))
    out = K.conv2d_fwd(_cl(x.view(N, Kd, 1, 1)), K.pack_w_fwd(w.view(O, Kd, 1, 1).cuda()), b.cuda(), 1, 1, relu=True,
                       out_dtype=torch.float32)
    assert (out.view(N, O).cpu() - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('cin,cout,k,s,p,hw', [(64, 64, 3, 1, 1, 20), (256, 128, 1, 2, 0, 17), (32, 36, 3, 2, 1, 9),
                                               (96, 256, 1, 1, 0, 7)])
def test_conv_tap_f32_parity_mode(cin, cout, k, s, p, hw, f32_contract):
    """loft_conv_tap_f32_v vs an fp64 CPU convolution: fp32 rounding only for the exact fp32 MFMA (1e-5 relative to the output
    scale); the split-bf16 contraction (16 mantissa bits per operand, fp32 sums) is held to the same bound."""
    import torch.nn.functional as F
    from bonai_amd import kernels as K
    torch.manual_seed(cin + cout + k)
    x = torch.randn(2, cin, hw, hw)
    w = torch.randn(cout, cin, k, k) * 0.05
    b = torch.randn(cout)
    res = torch.randn(2, cout, (hw + 2 * p - k) // s + 1, (hw + 2 * p - k) // s + 1)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), s, p) + res.double()).float()
    y = K.conv2d_fwd(x.cuda().contiguous(memory_format=torch.channels_last), K.pack_w_fwd(w.cuda(), torch.float32)[None],
                     b.cuda(), k, k, s, p, relu=True, residual=res.cuda().contiguous(memory_format=torch.channels_last),
                     out_dtype=torch.float32)
    assert y.dtype == torch.float32
    err = (y.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print(f'fp32 conv ({f32_contract}) {cin}->{cout} k{k}: max err / scale = {err:.2e}')
    assert err < 1e-5


@pytest.mark.parametrize('cin,cout,k,stride,xpad,opad', [(64, 64, 3, 1, None, None), (64, 192, 1, 1, None, None),
                                                         (32, 32, 3, 1, 64, 64), (32, 64, 3, 2, 64, None),
                                                         (64, 32, 1, 1, None, 64), (256, 64, 3, 1, None, None),
                                                         (64, 256, 1, 1, None, None)])
def test_narrow_channel_conv_fwd_bwd(cin, cout, k, stride, xpad, opad):
    """HRNet-sized convs (channels not multiples of 128): the 64-channel wgrad kernel, and 32-channel tensors carried in
    64-channel buffers through zero-padded weight packings (bonai_amd.nn.conv2d cout_pad / wider x)."""
    import torch.nn.functional as F
    from bonai_amd import nn as F2
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    B, H = 2, 20
    pad = k // 2
    x = torch.randn(B, cin, H, H, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).float()
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.relu(F.conv2d(xr, wr, br, stride, pad))
    gout = torch.randn(ref.shape, generator=g).to(torch.bfloat16).float()
    ref.backward(gout)
    xin = x
    if xpad:
        xin = torch.zeros(B, xpad, H, H)
        xin[:, :cin] = x
    xg = xin.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = F2.conv2d(xg, wg, bg, stride=stride, pad=pad, relu=True, cout_pad=opad)
    assert y.shape[1] == (opad or cout)
    if opad:
        assert y[:, cout:].abs().max().item() == 0
    scale = ref.abs().max().item()
    assert (y[:, :cout].float().cpu() - ref.detach()).abs().max().item() < 0.02 * scale
    gfull = torch.zeros(y.shape)
    gfull[:, :cout] = gout
    if opad:
        gfull[:, cout:] = 1.0       # gradient arriving on padded channels must not leak anywhere (y there is 0 -> relu masks it)
    y.backward(gfull.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    for name, got, want in (('x', xg.grad.float().cpu()[:, :cin], xr.grad), ('w', wg.grad.cpu(), wr.grad), ('b', bg.grad.cpu(), br.grad)):
        rel = (got - want).norm().item() / max(1e-6, want.norm().item())
        assert rel < 0.02, (name, rel)
    if xpad:
        assert xg.grad[:, cin:].abs().max().item() == 0


@pytest.mark.parametrize('N,Cin,Cout,hw,relu_in', [(37, 256, 1, 28, True), (513, 1024, 6, 1, True), (300, 1024, 2, 1, False),
                                                    (5, 64, 8, 7, True), (0, 256, 1, 28, True)])
def test_narrow_head_bwd_one_pass(N, Cin, Cout, hw, relu_in):
    """loft_narrow_head_bwd (mask logits / fc_cls+fc_reg / fc_offset backward in one pass over x) against the plain fp32
    formulas: gx = [x > 0] * g W, dW = g^T x, db = colsum(g)."""
    from bonai_amd import kernels as K
    torch.manual_seed(N + Cout)
    c4 = (Cout + 3) // 4 * 4
    x = torch.randn(N, Cin, hw, hw).relu().bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    g = torch.randn(N, c4, hw, hw).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin).cuda()
    gx, dw, db = K.narrow_head_bwd(g, x, w, relu_in=relu_in)
    assert gx.shape == x.shape and dw.shape == (Cout, Cin) and db.shape == (Cout,)
    if N == 0:
        assert float(dw.abs().sum()) == 0.0
        return
    x2 = x.float().permute(0, 2, 3, 1).reshape(-1, Cin)
    g2 = g.permute(0, 2, 3, 1).reshape(-1, c4)[:, :Cout]
    want_gx = g2 @ w
    if relu_in:
        want_gx = want_gx * (x2 > 0)
    got_gx = gx.float().permute(0, 2, 3, 1).reshape(-1, Cin)
    assert (got_gx - want_gx).abs().max().item() <= 1e-2 * max(1.0, want_gx.abs().max().item())     # bf16 output
    want_dw = g2.t() @ x2
    assert (dw - want_dw).abs().max().item() <= 1e-4 * max(1.0, want_dw.abs().max().item())
    assert (db - g2.sum(0)).abs().max().item() <= 1e-4 * max(1.0, g2.sum(0).abs().max().item())


@pytest.mark.parametrize('G,B,hw', [(4, 260, 7), (1, 300, 14), (2, 513, 7)])
def test_conv_pixel_major_tiles_on_roi_maps(G, B, hw):
    """Hundreds of small RoI maps (the FOA and mask heads): the 256x256 FAST kernel enumerates tile rows pixel-major and skips
    the taps that leave the map for a whole tile -- forward (bias + ReLU, fp32 and bf16 out) and data gradient (with a ReLU mask)
    against torch, per group, on inputs whose border behaviour matters (no zero padding in the data itself)."""
    from bonai_amd import kernels as K
    torch.manual_seed(G * 1000 + B)
    C = 256
    x = _r(torch.randn(G * B, C, hw, hw) + 0.5)
    w = _r(torch.randn(G, C, C, 3, 3) / (C * 9) ** 0.5)
    b = torch.randn(G, C)
    wp = torch.stack([K.pack_w_fwd(w[i].cuda()) for i in range(G)])
    out = K.conv2d_fwd(_cl(x), wp, b.cuda().contiguous(), 3, 3, 1, 1, relu=True, out_dtype=torch.float32, groups=G)
    outb = K.conv2d_fwd(_cl(x), wp, b.cuda().contiguous(), 3, 3, 1, 1, relu=True, groups=G)
    g = _r(torch.randn(G * B, C, hw, hw))
    wpt = torch.stack([K.pack_w_dgrad(w[i].cuda()) for i in range(G)])
    gx = K.conv2d_dgrad(_cl(g), wpt, (hw, hw), 3, 3, 1, 1, groups=G, mask=_cl(x))
    for i in range(G):
        sl = slice(i * B, (i + 1) * B)
        ref = F.relu(F.conv2d(x[sl], w[i], b[i], padding=1))
        assert (out[sl].cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
        assert (outb[sl].float().cpu() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
        refg = F.conv_transpose2d(g[sl], w[i], padding=1) * (x[sl] > 0)
        assert (gx[sl].float().cpu() - refg).abs().max().item() < 1e-2 * max(1.0, refg.abs().max().item())


@pytest.mark.parametrize('B,cin,cout,H,W,k', [(2, 64, 64, 192, 176, 3), (1, 32, 64, 260, 256, 3), (4, 64, 48, 128, 130, 3),
                                               (2, 64, 64, 200, 168, 1)])
def test_wgrad_patch_form_narrow_channels(B, cin, cout, H, W, k):
    """conv_wgrad64_patch_kernel (all taps of a narrow stride-1 conv from one staged 8x8 patch + halo): weight and bias gradient
    against torch, sizes that are not multiples of the patch, partial channel tiles, 3x3 and 1x1."""
    from bonai_amd import kernels as K
    torch.manual_seed(B * H + cout)
    pad = k // 2
    x = _r(torch.randn(B, cin, H, W))
    g = _r(torch.randn(B, cout, H, W))
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x, w, padding=pad).backward(g)
    dwp, db = K.conv2d_wgrad(_cl(g), _cl(x), k, k, 1, pad, with_bias=True)
    got = K.unpack_dw(dwp[0], w.shape).cpu()
    ref = w.grad
    assert (got - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())
    want_b = g.sum(dim=(0, 2, 3))
    assert (db[0, :cout].cpu() - want_b).abs().max().item() < 1e-3 * max(1.0, want_b.abs().max().item())


@pytest.mark.parametrize('B,H,W,k', [(2, 200, 168, 3), (1, 264, 250, 3), (5, 128, 128, 3), (2, 192, 176, 2)])
def test_conv64_patch_kernel(B, H, W, k):
    """conv64_patch_kernel (64 -> 64 channels, weights resident in LDS, taps fed from a staged patch halo): forward with bias +
    residual + ReLU and data gradient with a ReLU mask against torch; sizes that are not multiples of the 8 x 16 patch; k = 2
    exercises an asymmetric tap set (taps (0,0),(0,1),(1,0),(1,1))."""
    from bonai_amd import kernels as K
    torch.manual_seed(B * H + W)
    C = 64
    x = _r(torch.randn(B, C, H, W))
    res = _r(torch.randn(B, C, H, W))
    w = _r(torch.randn(C, C, k, k) / (C * k * k) ** 0.5)
    b = torch.randn(C)
    if k == 3:
        out = K.conv2d_fwd(_cl(x), K.pack_w_fwd(w.cuda())[None], b.cuda(), 3, 3, 1, 1, relu=True, residual=_cl(res))
        ref = F.relu(F.conv2d(x, w, b, padding=1) + res)
        assert (out.float().cpu() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
        g = _r(torch.randn(B, C, H, W))
        gx = K.conv2d_dgrad(_cl(g), K.pack_w_dgrad(w.cuda())[None], (H, W), 3, 3, 1, 1, mask=_cl(x))
        refg = F.conv_transpose2d(g, w, padding=1) * (x > 0)
        assert (gx.float().cpu() - refg).abs().max().item() < 1e-2 * max(1.0, refg.abs().max().item())
    else:
        taps = [(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)]
        wp = K.pack_w_fwd(w.cuda())
        out = K.empty_nhwc(B, C, H, W, torch.bfloat16, 'cuda')
        K.conv_tap(_cl(x), wp, out, B, H, W, C, C, H, W, H, W, taps, bias=b.cuda(), relu=False)
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b)
        assert (out.float().cpu() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('B,Cin,Cout,H,W,R,pad', [
    (300, 256, 256, 7, 7, 3, 1),        # 128-tile form
    (1300, 256, 256, 7, 7, 3, 1),       # 256-tile form (FOA head shape)
    (260, 128, 256, 14, 14, 3, 1),      # mask-head map
    (130, 128, 128, 5, 9, 3, 1),        # non-square map, 130 RoIs (K-steps straddle positions)
])
def test_wgrad_roi_maps_valid_rows_only(B, Cin, Cout, H, W, R, pad):
    """RoI-map form of conv_wgrad_kernel (K runs over the valid rectangle of each tap only, K-splits proportional to the tap's
    rows) vs torch-CPU autograd; LOFT_WGRAD_PIXMAJOR=0 is the plain enumeration."""
    from bonai_amd import kernels as K
    torch.manual_seed(11)
    x = _r(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, R, R, requires_grad=True)
    y = F.conv2d(x, w, None, stride=1, padding=pad)
    g = _r(torch.randn_like(y))
    (ref,) = torch.autograd.grad(y, w, g)
    tol = 1e-3 * max(1.0, ref.abs().max().item())
    for splits in (0, 1, 5):
        dwp = K.conv2d_wgrad(_cl(g), _cl(x), R, R, 1, pad, splits=splits)
        assert (K.unpack_dw(dwp[0], w.shape).cpu() - ref).abs().max().item() < tol, splits
    dwp, db = K.conv2d_wgrad(_cl(g), _cl(x), R, R, 1, pad, with_bias=True)
    want_b = g.sum(dim=(0, 2, 3))
    assert (db[0].cpu() - want_b).abs().max().item() < 2e-3 * max(1.0, want_b.abs().max().item())
    assert (K.unpack_dw(dwp[0], w.shape).cpu() - ref).abs().max().item() < tol


@pytest.mark.parametrize('cin,cout,k,s,p,hw,groups', [(64, 64, 3, 1, 1, 20, 1), (256, 128, 1, 2, 0, 17, 1), (32, 64, 3, 2, 1, 9, 1),
                                                      (96, 256, 1, 1, 0, 7, 1), (64, 64, 3, 1, 1, 7, 4)])
def test_fp32_parity_backward_kernels(cin, cout, k, s, p, hw, groups, f32_contract):
    """parity_f32.hip, the backward of the fp32 parity mode: loft_conv_wgrad_f32 (v_mfma_f32_32x32x2_f32, exact fp32 products) and
    the fp32 data gradient through loft_conv_tap_f32 against fp64 autograd of the same convolution; the fp32 ReLU-backward and
    the FPN adjoints (2x2 block sum, stride-2 scatter) against their definitions -- fp32 rounding only."""
    import torch.nn.functional as F
    from bonai_amd import kernels as K
    torch.manual_seed(cin + cout + k + groups)
    B = 2
    x = torch.randn(groups * B, cin, hw, hw, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(groups, cout, cin, k, k, dtype=torch.float64) * 0.05).requires_grad_(True)
    oh = (hw + 2 * p - k) // s + 1
    g = torch.randn(groups * B, cout, oh, oh, dtype=torch.float64)
    ys = [F.conv2d(x[i * B:(i + 1) * B], w[i], None, s, p) for i in range(groups)]
    torch.cat(ys).backward(g)
    xc = x.detach().float().cuda().contiguous(memory_format=torch.channels_last)
    gc = g.float().cuda().contiguous(memory_format=torch.channels_last)
    dwp, db = K.conv2d_wgrad(gc, xc, k, k, s, p, groups=groups, with_bias=True)
    assert dwp.dtype == torch.float32
    for i in range(groups):
        dw = K.unpack_dw(dwp[i], (cout, cin, k, k)).cpu()
        want = w.grad[i].float()
        print(f'fp32 wgrad ({f32_contract}) {cin}->{cout} k{k}: max err / scale = '
              f'{(dw - want).abs().max().item() / max(1.0, want.abs().max().item()):.2e}')
        assert (dw - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), (i, (dw - want).abs().max().item())
    want_b = g.view(groups, B, cout, -1).sum(dim=(1, 3)).float()
    assert (db[:, :cout].cpu() - want_b).abs().max().item() <= 2e-5 * max(1.0, want_b.abs().max().item())
    if groups == 1:
        wpt = w[0].detach().float().cuda().permute(2, 3, 1, 0).reshape(k * k, cin, cout).contiguous()[None]   # fp32 [T, Cin, Cout]
        gx = K.conv2d_dgrad(gc, wpt, (hw, hw), k, k, s, p, out_dtype=torch.float32)
        assert (gx.cpu() - x.grad.float()).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())
    # elementwise adjoints of the parity mode
    y = torch.randn(B, 16, 10, 12).cuda().contiguous(memory_format=torch.channels_last)
    gg = torch.randn_like(y)
    assert torch.equal(K.relu_bwd(gg, y), gg * (y > 0))
    fine = torch.randn(B, 16, 10, 12).cuda().contiguous(memory_format=torch.channels_last)
    coarse = torch.randn(B, 16, 5, 6).cuda().contiguous(memory_format=torch.channels_last)
    want = coarse + fine.view(B, 16, 5, 2, 6, 2).sum(dim=(3, 5))
    got = K.downsum2x_add_(coarse.clone(memory_format=torch.channels_last), fine)
    assert (got - want).abs().max().item() <= 1e-5
    big = torch.randn(B, 16, 9, 11).cuda().contiguous(memory_format=torch.channels_last)
    small = torch.randn(B, 16, 5, 6).cuda().contiguous(memory_format=torch.channels_last)
    want = big.clone()
    want[:, :, ::2, ::2] += small
    got = K.subsample2_adjoint_add_(big.clone(memory_format=torch.channels_last), small)
    assert (got - want).abs().max().item() <= 1e-6
