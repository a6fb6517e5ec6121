"""GPU: the binary16 build of the library (libloft_hip_f16.so; `fp16 = dict(loss_scale=512.)`, the reference's
mmdet/core/fp16/hooks.py:11-135 recipe: half activations / operands, fp32 master weights, fp32 loss islands, static loss scale).

The kernels are the same sources compiled with the 16-bit type switched (loft_common.h), so the checks are: (1) the MFMA conv
forward / data gradient / weight gradient against torch's fp32 conv on fp16-ROUNDED operands -- fp32 accumulation, so only the
output rounding (2^-11 relative) separates them; (2) a foreign 16-bit tensor is rejected, not reinterpreted; (3) the LOFT R50
training step in fp16 against the reference-made fixture, at a tighter tolerance than the bf16 path's (11 vs 8 mantissa bits);
(4) BASELINE config 5 as stated -- HRNetV2p-W32 in fp16 with loss scale 512 -- one full optimisation step against the CPU oracle.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture()
def fp16_mode():
    from bonai_amd import lib as L
    prev = L.set_act16(torch.float16)      # (cached frozen-layer packings are keyed by the 16-bit type: nothing to clear)
    yield
    L.set_act16(prev)


@pytest.mark.parametrize('shape', [(2, 256, 256, 24, 20, 3), (3, 128, 64, 17, 33, 1), (1, 64, 128, 40, 40, 3)])
def test_conv_fp16_vs_torch_on_rounded_operands(fp16_mode, shape):
    from bonai_amd import kernels as K
    B, Cin, Cout, H, W, R = shape
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, Cin, H, W, device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, R, R, device='cuda', generator=g) * (2.0 / (Cin * R * R)) ** 0.5).half()
    b = torch.randn(Cout, device='cuda', generator=g)
    pad = R // 2
    y = K.conv2d_fwd(x, K.pack_w_fwd(w.float())[None], b[None], R, R, 1, pad, relu=True)
    assert y.dtype == torch.float16
    want = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b, 1, pad))
    assert (y.float() - want).abs().max().item() <= 1e-3 * want.abs().max().item() + 1e-4
    go = torch.randn(B, Cout, H, W, device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    gx = K.conv2d_dgrad(go, K.pack_w_dgrad(w.float())[None], (H, W), R, R, 1, pad)
    wantx = torch.nn.grad.conv2d_input(x.shape, w.float(), go.float(), 1, pad)
    assert (gx.float() - wantx).abs().max().item() <= 1e-3 * wantx.abs().max().item() + 1e-4
    if Cin % 128 == 0 and Cout % 128 == 0 or (Cin <= 64 and Cout <= 64):
        pass
    dwp = K.conv2d_wgrad(go, x, R, R, 1, pad)
    dw = K.unpack_dw(dwp[0], w.shape)
    wantw = torch.nn.grad.conv2d_weight(x.float(), w.shape, go.float(), 1, pad)
    assert (dw - wantw).abs().max().item() <= 1e-3 * wantw.abs().max().item() + 1e-4


def test_foreign_16bit_type_is_rejected(fp16_mode):
    from bonai_amd import kernels as K, lib as L
    x = torch.randn(1, 128, 8, 8, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 128, 1, 1, device='cuda')
    with pytest.raises(L.LoftHipError):
        K.conv2d_fwd(x, K.pack_w_fwd(w)[None], None, 1, 1)
    rois = torch.tensor([[0, 1., 1., 20., 20.]], device='cuda')
    with pytest.raises(L.LoftHipError):
        K.roi_align_fwd([x.float().bfloat16()], rois, 7, [4])


def _build_r50():
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    return m.cuda().train()


def test_r50_step_fp16_vs_reference_fixture(fp16_mode):
    from bonai_amd.synth import make_batch
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    m = _build_r50()
    data = make_batch(batch, size, num_gt, device='cuda')
    feats = m.extract_feat(data['img'])
    assert all(f.dtype == torch.float16 for f in feats)
    for i, f in enumerate(feats):
        want = torch.from_numpy(gd[f'feat_{i}_crop'])
        got = f[:, :8, :6, :6].float().cpu()
        scale = float(gd[f'feat_{i}_absmean'])
        assert (got - want).abs().max().item() < 0.01 * scale * 4, (i, (got - want).abs().max().item(), scale)   # bf16 path: 0.03
    out = m.train_step(data)
    lv = dict(out['log_vars'].items())
    tol = dict(loss_rpn_cls=0.005, loss_rpn_bbox=0.01, loss_cls=0.01, loss_bbox=0.02, loss_mask=0.01, loss_offset=0.02, loss=0.02)
    for k, t in tol.items():
        want = float(gd['log_' + k])
        assert abs(lv[k] - want) <= t * max(1.0, abs(want)), (k, lv[k], want)
    (out['loss'] * 512.0).backward()          # the static loss scale of the recipe; gradients compared after un-scaling
    grads = {n: p.grad / 512.0 for n, p in m.named_parameters() if p.grad is not None}
    for k in gd.files:
        if k.startswith('gradnorm_'):
            n = k[len('gradnorm_'):]
            want = float(gd[k])
            got = float(grads[n].norm()) if n in grads else 0.0
            assert np.isfinite(got) and abs(got - want) <= 0.04 * max(1e-2, want), (n, got, want)   # bf16 path: 0.08


def test_config5_hrnet_fp16_loss_scale_512_step(fp16_mode):
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_hrnetv2p_w32_2x_bonai.py'))
    assert cfg.fp16['loss_scale'] == 512.0
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.cuda().train()
    data, cpu = make_batch(1, 256, 6, device='cuda'), make_batch(1, 256, 6)
    with torch.no_grad():
        ref = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'])
    tr = Trainer(m, lr=1e-3, loss_scale=cfg.fp16['loss_scale'])
    before = tr.arena.data.clone()
    out = tr.train_step(data)
    assert m.backbone.conv1.weight.dtype == torch.float32          # fp32 master weights
    for k, v in ref.items():
        if k.startswith('loss'):
            assert abs(out['log_vars'][k] - float(v.sum())) <= 0.03 * max(1.0, abs(float(v.sum()))), (k, out['log_vars'][k], float(v.sum()))
    assert torch.isfinite(tr.arena.grad).all() and torch.isfinite(tr.arena.data).all()
    assert float((tr.arena.data - before).abs().sum()) > 0
