"""GPU parity of the HRNet-W32 + HRFPN path (BASELINE config 5): resampling / fusion kernels vs torch CPU ops, the stem
kernel, and the whole backbone + neck vs the fixture produced by the REFERENCE's own hrnet.py / hrfpn.py
(tests/golden/hrnet_128.npz): fp32 parity mode at 1e-3, bf16 training path at bf16 tolerances incl. gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fuse_sum_relu_and_backward(dtype):
    from bonai_amd import nn as F2
    g = torch.Generator().manual_seed(1)
    B, C, H = 2, 64, 16
    terms = [torch.randn(B, C, H >> s, H >> s, generator=g).to(dtype).float() for s in (0, 1, 2, 0)]
    shifts = (0, 1, 2, 0)
    ref_in = [t.clone().requires_grad_(True) for t in terms]
    ref = F.relu(sum(F.interpolate(t, scale_factor=2 ** s, mode='nearest') if s else t for t, s in zip(ref_in, shifts)))
    gout = torch.randn(ref.shape, generator=g).to(dtype).float()
    ref.backward(gout)
    ins = [_nhwc(t.cuda().to(dtype)).requires_grad_(True) for t in terms]
    y = F2.fuse_sum_relu(ins, shifts)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (y.float().cpu() - ref.detach()).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    y.backward(_nhwc(gout.cuda().to(dtype)))
    for a, b in zip(ins, ref_in):
        assert (a.grad.float().cpu() - b.grad).abs().max().item() < tol * max(1.0, b.grad.abs().max().item())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_hrfpn_concat_and_avgpool(dtype):
    from bonai_amd import nn as F2
    g = torch.Generator().manual_seed(2)
    B, H = 2, 32
    chans = [64, 64, 128, 256]
    xs = [torch.randn(B, c, H >> i, H >> i, generator=g).to(dtype).float() for i, c in enumerate(chans)]
    rin = [x.clone().requires_grad_(True) for x in xs]
    ref = torch.cat([rin[0]] + [F.interpolate(x, scale_factor=2 ** i, mode='bilinear') for i, x in enumerate(rin) if i], 1)
    pooled = [F.avg_pool2d(ref, 2 ** i, 2 ** i) for i in (1, 2, 3)]
    gouts = [torch.randn(p.shape, generator=g).to(dtype).float() for p in [ref] + pooled]
    (ref * gouts[0]).sum().backward(retain_graph=True)
    for p, go in zip(pooled, gouts[1:]):
        (p * go).sum().backward(retain_graph=True)
    ins = [_nhwc(x.cuda().to(dtype)).requires_grad_(True) for x in xs]
    cat = F2.hrfpn_concat(ins)
    outs = [cat] + [F2.avgpool(cat, i) for i in (1, 2, 3)]
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for o, r in zip(outs, [ref] + pooled):
        assert (o.float().cpu() - r.detach()).abs().max().item() < tol * max(1.0, r.abs().max().item())
    loss = sum((o.float() * _nhwc(go.cuda())).sum() for o, go in zip(outs, gouts))
    loss.backward()
    for a, b in zip(ins, rin):
        assert (a.grad.float().cpu() - b.grad).abs().max().item() < (1e-4 if dtype == torch.float32 else 4e-2) * max(1.0, b.grad.abs().max().item())


def test_stem3x3s2_fwd_bwd():
    from bonai_amd import nn as F2
    from bonai_amd.loft.backbone import FrozenStatBN
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 70, 52
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.3
    bn = FrozenStatBN(64)
    bn.weight.data = torch.rand(64, generator=g) + 0.5
    bn.bias.data = torch.randn(64, generator=g) * 0.1
    bn.running_mean.data = torch.randn(64, generator=g) * 0.1
    bn.running_var.data = torch.rand(64, generator=g) + 0.5
    wr = w.clone().requires_grad_(True)
    gam, bet = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    ref = F.relu(F.batch_norm(F.conv2d(img, wr, None, 2, 1), bn.running_mean, bn.running_var, gam, bet, False, 0.0, bn.eps))
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    bn = bn.cuda()
    wg = w.cuda().requires_grad_(True)
    y = F2.stem3x3s2(img.cuda(), wg, bn, torch.float32)
    assert (y.cpu() - ref.detach()).abs().max().item() < 1e-4
    y.backward(_nhwc(gout.cuda()))
    for name, got, want in (('w', wg.grad.cpu(), wr.grad), ('gamma', bn.weight.grad.cpu(), gam.grad), ('beta', bn.bias.grad.cpu(), bet.grad)):
        assert (got - want).abs().max().item() < 2e-4 * max(1.0, want.abs().max().item()), name


def _build():
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from oracle.synth_weights import synth_tensor
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_hrnetv2p_w32_2x_bonai.py'))
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    return m.cuda().train()


def test_hrnet_hrfpn_vs_reference_fixture():
    from bonai_amd.synth import make_batch
    gd = np.load(os.path.join(GOLD, 'hrnet_128.npz'))
    size = int(gd['meta'][0])
    m = _build()
    img = make_batch(1, size, 4, device='cuda')['img']
    # fp32 parity mode: forward at the north-star tolerance
    m.backbone.compute_dtype = torch.float32
    with torch.no_grad():
        ys = m.backbone(img)
        outs = m.neck(ys)
    for i, y in enumerate(ys):
        c = int(gd[f'bb_{i}_shape'][1])
        assert y.dtype == torch.float32 and tuple(y.shape[2:]) == tuple(gd[f'bb_{i}_shape'][2:])
        if y.shape[1] > c:
            assert y[:, c:].abs().max().item() == 0           # the padded half of the 32-channel branch stays exactly zero
        assert (y[:, :8, :6, :6].cpu() - torch.from_numpy(gd[f'bb_{i}_crop'])).abs().max().item() < 1e-3 * float(gd[f'bb_{i}_absmean'])
        assert abs(float(y.double().sum()) - float(gd[f'bb_{i}_sum'])) <= 1e-4 * y[:, :c].numel() * float(gd[f'bb_{i}_absmean'])
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(gd[f'neck_{i}_shape'])
        assert (o[:, :8, :6, :6].cpu() - torch.from_numpy(gd[f'neck_{i}_crop'])).abs().max().item() < 1e-3 * float(gd[f'neck_{i}_absmean'])
    # bf16 training path: outputs, then the gradients of the fixture's scalar
    m.backbone.compute_dtype = torch.bfloat16
    outs = m.neck(m.backbone(img))
    for i, o in enumerate(outs):
        assert o.dtype == torch.bfloat16
        assert (o[:, :8, :6, :6].float().cpu() - torch.from_numpy(gd[f'neck_{i}_crop'])).abs().max().item() < 0.06 * float(gd[f'neck_{i}_absmean'])
    loss = sum((o.float() * torch.linspace(-1, 1, o.numel(), device='cuda').view(o.shape)).sum() for o in outs) / 1000.0
    assert abs(float(loss) - float(gd['loss'])) <= 0.05 * abs(float(gd['loss'])) + 0.5
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for k in gd.files:
        if k.startswith('gradnorm_'):
            n = k[len('gradnorm_'):]
            want = float(gd[k])
            got = float(grads[n].norm())
            assert abs(got - want) <= 0.1 * max(1e-3, want), (n, got, want)


def test_config5_train_step_runs():
    """LOFT + FOA on HRNet-W32 / HRFPN: one bf16 training step on a 256^2 tile; losses vs the CPU oracle."""
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    from oracle.synth_weights import synth_tensor
    RandomSampler.choice_mode = 'first'
    m = _build()
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    data = make_batch(1, 256, 6, device='cuda')
    cpu = make_batch(1, 256, 6)
    with torch.no_grad():
        ref = M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'])
        m.backbone.compute_dtype = torch.float32
        lv = dict(m.train_step(data)['log_vars'].items())
    for k, v in ref.items():
        if k.startswith('loss'):
            assert abs(lv[k] - float(v.sum())) <= 1e-3 * max(1.0, abs(float(v.sum()))), (k, lv[k], float(v.sum()))
    m.backbone.compute_dtype = torch.bfloat16
    out = m.train_step(data)
    for k, v in ref.items():
        if k.startswith('loss'):
            assert abs(out['log_vars'][k] - float(v.sum())) <= 0.08 * max(1.0, abs(float(v.sum()))), (k, out['log_vars'][k], float(v.sum()))
    out['loss'].backward()
    for n, p in m.named_parameters():
        if n.startswith('backbone'):
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_hrnet_branch_streams_and_graph_replay():
    """HRModule branches on their own HIP streams (loft.hrnet.BRANCH_STREAMS) reproduce the single-stream forward and backward
    bit for bit where sums are ordered (features) and within atomics noise (gradients); and the whole backbone + HRFPN section
    replayed from hipGraphs (Trainer(graph_features=True): branch streams and the weight-gradient stream forked inside the
    capture, one zeroed slab for the accumulation buffers) gives the eager Trainer's gradients."""
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import hrnet as H
    from bonai_amd.loft.core import RandomSampler
    from bonai_amd.synth import make_batch
    RandomSampler.choice_mode = 'first'
    try:
        data = make_batch(2, 256, 6, device='cuda')
        m = _build()
        with torch.no_grad():
            ref = [f.clone() for f in m.extract_feat(data['img'])]
        H.BRANCH_STREAMS = [torch.cuda.Stream() for _ in range(3)]
        try:
            with torch.no_grad():
                got = m.extract_feat(data['img'])
            torch.cuda.synchronize()
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
        finally:
            H.BRANCH_STREAMS = None
        eager = Trainer(m, lr=0.0, momentum=0.0, weight_decay=0.0)
        eager.train_step(data, lr=0.0)
        torch.cuda.synchronize()
        want = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad}
        m2 = _build()
        tr = Trainer(m2, lr=0.0, momentum=0.0, weight_decay=0.0, graph_features=True)
        for step in range(4):
            tr.train_step(data, lr=0.0)
            torch.cuda.synchronize()
            assert tr._fgraphs.failed is None, tr._fgraphs.failed
            bad = []
            for n, p in m2.named_parameters():
                if p.requires_grad:
                    d, s = (p.grad - want[n]).norm().item(), want[n].norm().item()
                    if d > 2e-2 * s + 1e-6:
                        bad.append((n, d, s))
            assert not bad, (step, len(bad), bad[:4])
        assert tr._fgraphs.ready and H.BRANCH_STREAMS is None
    finally:
        RandomSampler.choice_mode = 'random'
