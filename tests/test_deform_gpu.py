"""GPU parity of the DCNv2 path (BASELINE config 4): loft_mdcn_sample_fwd/bwd and the composed
modulated_deform_conv2d vs the CPU oracle (oracle/ops_ref.py: mdcn_im2col / mdcn_pack, parity unpinned at the mmcv
boundary -- see the oracle header) and its torch-autograd gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def _case(seed, B, C, H, W, kh, stride, pad, DG, off_scale=1.5):
    g = torch.Generator().manual_seed(seed)
    K = kh * kh
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    x = torch.randn(B, C, H, W, generator=g)
    raw = torch.randn(B, 3 * DG * K, OH, OW, generator=g)
    raw[:, :2 * DG * K] *= off_scale
    raw[0, :2 * DG * K, 0, 0] = 40.0            # far outside the map -> zero sample, zero gradients
    raw[0, 0, 1, 1] = -1.0 + 1e-3               # just inside the -1 border
    return x, raw, (OH, OW)


def _oracle_cols(x, raw, kh, stride, pad, DG):
    K = kh * kh
    off, mask = raw[:, :2 * DG * K], torch.sigmoid(raw[:, 2 * DG * K:3 * DG * K])
    return R.mdcn_im2col(x, off, mask, kh, kh, stride, pad, 1, DG)          # [B,C,K,OH,OW]


@pytest.mark.parametrize('B,C,H,W,kh,stride,pad,DG', [(2, 64, 13, 11, 3, 1, 1, 1), (1, 128, 16, 16, 3, 2, 1, 1),
                                                      (2, 256, 9, 9, 1, 1, 0, 1), (1, 64, 10, 12, 3, 1, 1, 2),
                                                      (1, 1024, 6, 6, 1, 1, 0, 1)])
def test_sample_fwd_bwd_fp32(B, C, H, W, kh, stride, pad, DG):
    from bonai_amd import kernels as K
    x, raw, (OH, OW) = _case(B * C + H, B, C, H, W, kh, stride, pad, DG)
    KK = kh * kh
    omc = (3 * DG * KK + 3) // 4 * 4
    om = torch.zeros(B, omc, OH, OW)
    om[:, :3 * DG * KK] = raw
    xr, rawr = x.clone().requires_grad_(True), raw.clone().requires_grad_(True)
    ref = _oracle_cols(xr, rawr, kh, stride, pad, DG)
    got = K.mdcn_sample_fwd(_nhwc(x.cuda()), _nhwc(om.cuda()), kh, kh, stride, pad, 1, DG)
    got5 = got.cpu().permute(0, 2, 3, 1).reshape(B, OH, OW, KK, C).permute(0, 4, 3, 1, 2)
    assert (got5 - ref.detach()).abs().max().item() < 1e-5
    gcol = torch.randn(ref.shape, generator=torch.Generator().manual_seed(7))
    ref.backward(gcol)
    dcol = _nhwc(gcol.permute(0, 2, 1, 3, 4).reshape(B, KK * C, OH, OW).cuda())
    dx, dom = K.mdcn_sample_bwd(_nhwc(x.cuda()), _nhwc(om.cuda()), dcol, kh, kh, stride, pad, 1, DG)
    assert (dx.cpu() - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item())
    dref = rawr.grad
    assert (dom.cpu()[:, :3 * DG * KK] - dref).abs().max().item() < 2e-4 * max(1.0, dref.abs().max().item())
    assert dom.cpu()[:, 3 * DG * KK:].abs().max().item() == 0 if omc > 3 * DG * KK else True


def test_sample_bf16_matches_fp32_of_rounded_inputs():
    from bonai_amd import kernels as K
    x, raw, (OH, OW) = _case(5, 2, 128, 12, 12, 3, 1, 1, 1)
    xb = x.to(torch.bfloat16)
    om = torch.zeros(2, 28, OH, OW)
    om[:, :27] = raw
    ref = _oracle_cols(xb.float(), raw, 3, 1, 1, 1)
    got = K.mdcn_sample_fwd(_nhwc(xb.cuda()), _nhwc(om.cuda()), 3, 3, 1, 1, 1, 1)
    assert got.dtype == torch.bfloat16
    got5 = got.float().cpu().permute(0, 2, 3, 1).reshape(2, OH, OW, 9, 128).permute(0, 4, 3, 1, 2)
    assert (got5 - ref).abs().max().item() < 2 ** -8 * max(1.0, ref.abs().max().item())      # one bf16 rounding


@pytest.mark.parametrize('cin,cout,k,stride', [(128, 128, 3, 1), (128, 128, 3, 2), (256, 256, 1, 1)])
def test_modulated_deform_conv2d_vs_oracle(cin, cout, k, stride):
    """Composed op (conv_offset -> sample -> MFMA contraction), forward in both modes and bf16 gradients."""
    from bonai_amd import nn as F2
    g = torch.Generator().manual_seed(cin + k + stride)
    B, H = 2, 14
    pad = k // 2
    x = torch.randn(B, cin, H, H, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k)) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    w_off = torch.randn(3 * k * k, cin, k, k, generator=g) * (0.5 / (cin * k * k)) ** 0.5
    b_off = torch.randn(3 * k * k, generator=g) * 0.1
    ps = [t.clone().requires_grad_(True) for t in (x, w, b, w_off, b_off)]
    ref = R.mdcn_pack(ps[0], ps[1], ps[2], ps[3], ps[4], stride, pad)
    # fp32 parity mode, forward
    with torch.no_grad():
        y32 = F2.modulated_deform_conv2d(_nhwc(x.cuda()), w.cuda(), b.cuda(), w_off.cuda(), b_off.cuda(), stride, pad)
    assert y32.dtype == torch.float32
    assert (y32.cpu() - ref.detach()).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    # bf16 training path
    gs = [t.clone().cuda().requires_grad_(True) for t in (w, b, w_off, b_off)]
    xb = _nhwc(x.cuda().to(torch.bfloat16)).requires_grad_(True)
    y = F2.modulated_deform_conv2d(xb, gs[0], gs[1], gs[2], gs[3], stride, pad)
    scale = ref.abs().max().item()
    assert (y.float().cpu() - ref.detach()).abs().max().item() < 0.03 * scale
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    y.backward(_nhwc(gout.cuda().to(torch.bfloat16)))
    for name, got, want in (('x', xb.grad.float().cpu(), ps[0].grad), ('w', gs[0].grad.cpu(), ps[1].grad),
                            ('b', gs[1].grad.cpu(), ps[2].grad), ('w_off', gs[2].grad.cpu(), ps[3].grad),
                            ('b_off', gs[3].grad.cpu(), ps[4].grad)):
        rel = (got - want).norm().item() / max(1e-6, want.norm().item())
        assert rel < 0.05, (name, rel)


def _build(cfg_name):
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', cfg_name))
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    return m.cuda().train(), sd


def test_config4_features_and_losses_vs_oracle():
    """LOFT + DCNv2 (backbone c3-c5 + neck) on a 256^2 tile: fp32 parity mode features/losses vs the CPU oracle at 1e-3,
    then one bf16 training step (finite losses, gradients reach every conv_offset)."""
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    m, sd = _build('loft_foa_r50_fpn_mdconv_c3-c5_2x_bonai.py')
    data = make_batch(1, 256, 6, device='cuda')
    cpu = make_batch(1, 256, 6)
    with torch.no_grad():
        want = M.fpn(sd, M.backbone(sd, cpu['img']))
        ref_losses = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'])
        m.backbone.compute_dtype = torch.float32
        feats = m.extract_feat(data['img'])
        for f, r in zip(feats, want):
            err = (f.cpu() - r).abs()          # 13 stacked data-dependent samplers amplify fp32 rounding: 1e-3 of the range
            assert err.max().item() < 1e-3 * r.abs().max().item() and err.mean().item() < 1e-4 * r.abs().mean().item()
        lv = dict(m.train_step(data)['log_vars'].items())
        for k, v in ref_losses.items():
            if k.startswith('loss'):
                assert abs(lv[k] - float(v.sum())) <= 1e-3 * max(1.0, abs(float(v.sum()))), (k, lv[k], float(v.sum()))
    m.backbone.compute_dtype = torch.bfloat16
    out = m.train_step(data)
    assert np.isfinite(float(out['loss']))
    for k, v in ref_losses.items():
        if k.startswith('loss'):
            assert abs(out['log_vars'][k] - float(v.sum())) <= 0.08 * max(1.0, abs(float(v.sum()))), (k, out['log_vars'][k], float(v.sum()))
    out['loss'].backward()
    for n, p in m.named_parameters():
        if 'conv_offset' in n:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
            if n.startswith('backbone') or 'convs.0.' in n:     # (coarse neck levels may see no sampled anchor / RoI on a 256^2 tile)
                assert p.grad.abs().sum().item() > 0, n
