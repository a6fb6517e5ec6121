"""GPU parity of loft_bneck_pair_bf16 (bneck_pair.hip): the end of bottleneck k + the start of bottleneck k+1 in one launch, forward and
backward, against (a) the separate launches of the library and (b) torch fp32 on the same 16-bit-rounded operands
(mmdet/models/backbones/resnet.py:266-298).  The fused launch keeps the separate launches' rounding points and fp32 operation order; the
residual (and the bias) are added on the matrix pipe as exact x 1.0 products, whose final rounding differs from the VALU add once in
~5e5 elements (measured: 0 of 655 360 at 256 planes, 1 of 524 288 at 128): (a) is "the same bits but for a handful of elements"."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t, dtype=torch.bfloat16):
    return t.to('cuda', dtype).contiguous(memory_format=torch.channels_last)


def _rows(t):
    """[B,C,H,W] channels_last -> fp32 [B*H*W, C]"""
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).float()


def _close_to_ulp(got, want, frac_ok=1e-4, exact=False):
    """got / want: bf16 tensors that went through the same rounding points; equal up to fp32 summation order = identical on nearly
    every element, one bf16 unit apart on a few (a sum that lands next to a rounding boundary).  exact: bit for bit."""
    g, w = got.float(), want.float()
    if exact:
        frac_ok = max(2.0 / g.numel(), 4e-6)
    diff = (g - w).abs()
    # (+ an absolute floor: an output that is the difference of O(1) partial sums carries their fp32 rounding, ~1e-6, whatever its size)
    ulp = torch.maximum(g.abs(), w.abs()) * 2.0 ** -7 + 1e-5 * float(w.abs().max())
    assert (diff <= 1.01 * ulp).all(), float((diff / ulp).max())
    assert (diff > 0).float().mean().item() <= frac_ok, (diff > 0).float().mean().item()


CASES = [(2, 128, 16, 32), (1, 256, 16, 16), (2, 256, 8, 24), (1, 128, 8, 16)]      # B, P, H, W  (B*H*W % 128 == 0)


@pytest.mark.parametrize('B,P,H,W', CASES)
def test_pair_forward_matches_separate_launches_and_fp32(B, P, H, W):
    from bonai_amd import kernels as K
    torch.manual_seed(5)
    C = 4 * P
    t2 = _cl(torch.randn(B, P, H, W).relu())
    x = _cl(torch.randn(B, C, H, W).relu())
    w3 = torch.randn(C, P, 1, 1, device='cuda') / P ** 0.5
    w1n = torch.randn(P, C, 1, 1, device='cuda') / C ** 0.5
    b3, b1n = torch.randn(C, device='cuda') * 0.1, torch.randn(P, device='cuda') * 0.1
    wp3, wp1n = K.pack_w_fwd(w3), K.pack_w_fwd(w1n)
    out_s = K.conv2d_fwd(t2, wp3[None], b3[None], 1, 1, 1, 0, relu=True, residual=x)
    t1_s = K.conv2d_fwd(out_s, wp1n[None], b1n[None], 1, 1, 1, 0, relu=True)
    assert K.bneck_pair_ok(t2, C)
    (k3, k1), _ = K.pack_k8([wp3, wp1n])
    out_f, t1_f = K.bneck_pair(t2, k3, b3, x, k1, b1n)
    torch.cuda.synchronize()
    assert out_f.shape == out_s.shape and t1_f.shape == t1_s.shape
    assert out_f.is_contiguous(memory_format=torch.channels_last) and t1_f.is_contiguous(memory_format=torch.channels_last)
    _close_to_ulp(out_f, out_s, exact=P == 256)
    _close_to_ulp(t1_f, t1_s, exact=P == 256)
    # the second product reads the fused launch's own (rounded) mid: compare it with a separate launch on THAT tensor
    _close_to_ulp(t1_f, K.conv2d_fwd(out_f, wp1n[None], b1n[None], 1, 1, 1, 0, relu=True), exact=True)
    # fp32 reference on the same rounded operands
    ref_out = (_rows(t2) @ wp3[0].float().t() + b3 + _rows(x)).relu()
    assert (_rows(out_f) - ref_out).abs().max().item() < 1e-2 * max(1.0, ref_out.abs().max().item())
    ref_t1 = (_rows(out_f) @ wp1n[0].float().t() + b1n).relu()
    assert (_rows(t1_f) - ref_t1).abs().max().item() < 1e-2 * max(1.0, ref_t1.abs().max().item())


@pytest.mark.parametrize('B,P,H,W', CASES)
def test_pair_backward_matches_separate_launches_and_fp32(B, P, H, W):
    from bonai_amd import kernels as K
    torch.manual_seed(6)
    C = 4 * P
    g_t1 = _cl(torch.randn(B, P, H, W))                 # gradient of t1_{k+1}
    g_sc = _cl(torch.randn(B, C, H, W))                 # gradient over block k+1's identity shortcut
    out_k = _cl(torch.randn(B, C, H, W).relu())         # ReLU output of block k (mask 1)
    t2 = _cl(torch.randn(B, P, H, W).relu())            # ReLU output of conv2 of block k (mask 2)
    w1n = torch.randn(P, C, 1, 1, device='cuda') / C ** 0.5
    w3 = torch.randn(C, P, 1, 1, device='cuda') / P ** 0.5
    wpt1n, wpt3 = K.pack_w_dgrad(w1n), K.pack_w_dgrad(w3)       # [1, C, P], [1, P, C]
    gx_s = K.conv2d_dgrad(g_t1, wpt1n[None], (H, W), 1, 1, 1, 0, residual=g_sc, mask=out_k)
    gt2_s = K.conv2d_dgrad(gx_s, wpt3[None], (H, W), 1, 1, 1, 0, mask=t2)
    (k1, k3), _ = K.pack_k8([wpt1n, wpt3])
    gx_f, gt2_f = K.bneck_pair(g_t1, k1, None, g_sc, k3, None, mask1=out_k, mask2=t2)
    torch.cuda.synchronize()
    _close_to_ulp(gx_f, gx_s, exact=P == 256)
    _close_to_ulp(gt2_f, gt2_s, exact=P == 256)
    _close_to_ulp(gt2_f, K.conv2d_dgrad(gx_f, wpt3[None], (H, W), 1, 1, 1, 0, mask=t2), exact=True)
    ref_gx = (_rows(g_t1) @ wpt1n[0].float().t() + _rows(g_sc)) * (_rows(out_k) > 0)
    assert (_rows(gx_f) - ref_gx).abs().max().item() < 1e-2 * max(1.0, ref_gx.abs().max().item())
    ref_gt2 = (_rows(gx_f) @ wpt3[0].float().t()) * (_rows(t2) > 0)
    assert (_rows(gt2_f) - ref_gt2).abs().max().item() < 1e-2 * max(1.0, ref_gt2.abs().max().item())
    assert (_rows(gx_f)[_rows(out_k) <= 0] == 0).all() and (_rows(gt2_f)[_rows(t2) <= 0] == 0).all()


def test_pair_rejects_what_it_does_not_serve():
    from bonai_amd import kernels as K
    from bonai_amd import lib as L
    t2 = _cl(torch.randn(1, 64, 16, 8))
    assert not K.bneck_pair_ok(t2, 256)                                    # 64 planes
    assert not K.bneck_pair_ok(_cl(torch.randn(1, 128, 9, 9)), 512)        # 81 rows
    x = _cl(torch.randn(1, 512, 16, 8))
    a = _cl(torch.randn(1, 128, 16, 8))
    w1, w2 = torch.zeros(16, 512, 8, device='cuda', dtype=torch.bfloat16), torch.zeros(64, 128, 8, device='cuda', dtype=torch.bfloat16)
    with pytest.raises(L.LoftHipError):                                    # forward form without its biases
        K.bneck_pair(a, w1, None, x, w2, None)
