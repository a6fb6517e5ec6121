"""Generate tests/golden/*.npz by running the REFERENCE's own python (imported from /root/reference with
the stand-in mmcv of mmcv_stub.py).  Runs only in the build container; the fixtures (inputs + expected
outputs, a few hundred KB) are committed, the reference is not.

    python oracle/ref_harness/make_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import mmcv_stub  # noqa: E402

mmcv_stub.install()
GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)


def T(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def core_ops():
    from mmdet.core.anchor import AnchorGenerator
    from mmdet.core.bbox.assigners import MaxIoUAssigner
    from mmdet.core.bbox.coder import DeltaXYWHBBoxCoder
    from mmdet.core.bbox.coder.delta_xy_offset_coder import DeltaXYOffsetCoder
    from mmdet.core.bbox.iou_calculators import bbox_overlaps
    from mmdet.models.losses import CrossEntropyLoss, L1Loss, SmoothL1Loss, accuracy
    from mmdet.models.roi_heads.attribute_heads.offset_head_expand_feature import OffsetHeadExpandFeature
    out = {}
    rng = np.random.RandomState(0)
    ag = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    for i, a in enumerate(ag.grid_anchors(sizes, device='cpu')):
        out[f'anchors_{i}'] = T(a)
    out['anchor_sizes'] = np.array(sizes)

    def boxes(n, size=512.):
        cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
        w, h = rng.uniform(4, 150, n), rng.uniform(4, 150, n)
        return torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size), dtype=torch.float32)
    rois, gts = boxes(200), boxes(200)
    deltas = torch.tensor(rng.randn(200, 4), dtype=torch.float32)
    deltas[0] = torch.tensor([0., 0., 30., -30.])
    out['rois'], out['gts'], out['deltas'] = T(rois), T(gts), T(deltas)
    for tag, stds in (('rpn', (1., 1., 1., 1.)), ('rcnn', (.1, .1, .2, .2))):
        c = DeltaXYWHBBoxCoder(target_stds=stds)
        out[f'decode_{tag}'] = T(c.decode(rois, deltas, max_shape=(512, 512, 3)))
        out[f'encode_{tag}'] = T(c.encode(rois, gts))
    oc = DeltaXYOffsetCoder()
    offs = torch.tensor(rng.uniform(-40, 40, (200, 2)), dtype=torch.float32)
    out['offs'] = T(offs)
    out['offset_encode'] = T(oc.encode(rois, offs))
    out['offset_decode'] = T(oc.decode(rois, deltas[:, :2], max_shape=[512, 512]))
    # IoU + assignment (both threshold sets of the LOFT config)
    ab, ag_ = boxes(3000), boxes(40)
    ab[:40] = ag_
    ab[100] = ab[101]
    out['assign_boxes'], out['assign_gts'] = T(ab), T(ag_)
    out['iou'] = T(bbox_overlaps(ag_, ab))
    for tag, (p, n, m) in (('rpn', (0.7, 0.3, 0.3)), ('rcnn', (0.5, 0.5, 0.5))):
        r = MaxIoUAssigner(p, n, min_pos_iou=m, match_low_quality=True, ignore_iof_thr=-1).assign(ab, ag_)
        out[f'assign_{tag}_gt_inds'], out[f'assign_{tag}_max'] = T(r.gt_inds), T(r.max_overlaps)
    # FOA pieces
    head = OffsetHeadExpandFeature(num_convs=1, share_expand_fc=True, expand_feature_num=4,
                                   loss_offset=dict(type='SmoothL1Loss', loss_weight=16.0))
    feat = torch.tensor(rng.randn(3, 4, 7, 7), dtype=torch.float32)
    out['foa_feat'] = T(feat)
    for k in range(4):
        out[f'foa_rot_{k}'] = T(head.expand_feature(feat, k))

    class _Res:
        pass
    res = []
    pos_list, ind_list, off_list = [], [], []
    for i in range(2):
        r = _Res()
        r.pos_bboxes = boxes(7 + i)
        r.pos_assigned_gt_inds = torch.tensor(rng.randint(0, 5, 7 + i))
        res.append(r)
        off_list.append(torch.tensor(rng.uniform(-30, 30, (5, 2)), dtype=torch.float32))
        pos_list.append(r.pos_bboxes)
        ind_list.append(r.pos_assigned_gt_inds)
    off_list[0][0] = torch.tensor([3.0, 0.0])   # axis-aligned offset: polar round trip leaves ~1e-16 residue
    tg = head.get_targets(res, off_list, None)
    for i in range(2):
        out[f'foa_pos_{i}'], out[f'foa_ind_{i}'], out[f'foa_gtoff_{i}'] = T(pos_list[i]), T(ind_list[i]), T(off_list[i])
    out['foa_targets'] = T(tg)
    pred = torch.tensor(rng.randn(4 * 15, 2), dtype=torch.float32)
    pred[0] = 0.
    det = boxes(15)
    out['foa_pred'], out['foa_det'] = T(pred), T(det)
    out['foa_fused'] = T(head.offset_fusion(pred))
    out['foa_offsets'] = head.get_offsets(pred, det, None, False)
    out['foa_loss'] = T(head.loss(pred, torch.tensor(rng.randn(60, 2), dtype=torch.float32) * 0 + 0.3)['loss_offset'])
    # losses (in-tree known answers live in tests/test_losses; here random inputs)
    logits = torch.tensor(rng.randn(50, 2), dtype=torch.float32)
    lab = torch.tensor(rng.randint(0, 2, 50))
    w = torch.tensor(rng.rand(50) > 0.3, dtype=torch.float32)
    out['l_logits'], out['l_lab'], out['l_w'] = T(logits), T(lab), T(w)
    out['l_ce'] = T(CrossEntropyLoss()(logits, lab, w, avg_factor=37.0))
    out['l_bce'] = T(CrossEntropyLoss(use_sigmoid=True)(logits[:, :1], lab, w, avg_factor=37.0))
    out['l_acc'] = T(accuracy(logits, lab))
    a, b = torch.tensor(rng.randn(50, 4), dtype=torch.float32) * 2, torch.tensor(rng.randn(50, 4), dtype=torch.float32)
    out['l_a'], out['l_b'] = T(a), T(b)
    out['l_l1'] = T(L1Loss()(a, b, w[:, None].expand(50, 4), avg_factor=50.0))
    out['l_sl1'] = T(SmoothL1Loss(loss_weight=16.0)(a, b))
    mp = torch.tensor(rng.randn(6, 1, 28, 28), dtype=torch.float32)
    mt = torch.tensor(rng.rand(6, 28, 28) > 0.5, dtype=torch.float32)
    out['l_mp'], out['l_mt'] = T(mp), T(mt)
    out['l_mask'] = T(CrossEntropyLoss(use_mask=True)(mp, mt, torch.zeros(6, dtype=torch.long)))
    np.savez_compressed(os.path.join(GOLD, 'core_ops.npz'), **out)
    print('core_ops.npz', len(out), 'arrays')


def e2e(size=256, batch=2, num_gt=10):
    """Full reference LOFT forward_train (+ backward) on a seeded tile with name-keyed synthetic weights and the
    deterministic 'first-k' sampling rule injected into RandomSampler."""
    from bonai_amd.config import Config
    from bonai_amd.synth import make_batch
    from mmdet.core import BitmapMasks
    from mmdet.core.bbox.samplers import RandomSampler
    from mmdet.models import build_detector
    from oracle.synth_weights import synth_state_dict
    RandomSampler.random_choice = lambda self, gallery, num: gallery[:num]
    cfg = Config.fromfile('/root/reference/configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py')
    cfg.model.pretrained = None
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict(synth_state_dict(m.state_dict()))
    m.train()
    data = make_batch(batch, size, num_gt)
    data['gt_masks'] = [BitmapMasks(x.numpy(), size, size) for x in data['gt_masks']]
    # capture intermediates
    cap = {}
    m.neck.register_forward_hook(lambda mod, i, o: cap.__setitem__('feats', o))
    m.rpn_head.register_forward_hook(lambda mod, i, o: cap.__setitem__('rpn', o))
    orig = m.roi_head.forward_train

    def spy(x, img_metas, proposal_list, *a, **k):
        cap['proposals'] = proposal_list
        return orig(x, img_metas, proposal_list, *a, **k)
    m.roi_head.forward_train = spy
    losses = m(**data)
    loss, log_vars = m._parse_losses(losses)
    loss.backward()
    out = {f'log_{k}': np.float32(v) for k, v in log_vars.items()}
    for i, f in enumerate(cap['feats']):
        out[f'feat_{i}_crop'] = T(f[:, :8, :6, :6])
        out[f'feat_{i}_absmean'] = np.float32(f.abs().mean().item())
    out['rpn_cls0_crop'] = T(cap['rpn'][0][0][:, :, :6, :6])
    out['rpn_reg4_crop'] = T(cap['rpn'][1][4][:, :, :2, :2])
    for i, p in enumerate(cap['proposals']):
        out[f'proposals_{i}_n'] = np.int64(p.shape[0])
        out[f'proposals_{i}_head'] = T(p[:32])
    gr = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for n in ['backbone.layer2.0.conv1.weight', 'backbone.layer4.2.bn3.weight', 'backbone.layer4.2.bn3.bias',
              'neck.lateral_convs.0.conv.weight', 'neck.fpn_convs.3.conv.bias', 'rpn_head.rpn_conv.weight',
              'rpn_head.rpn_reg.bias', 'roi_head.bbox_head.shared_fcs.0.weight', 'roi_head.bbox_head.fc_reg.weight',
              'roi_head.mask_head.upsample.weight', 'roi_head.mask_head.conv_logits.weight',
              'roi_head.offset_head.expand_convs.2.0.weight', 'roi_head.offset_head.expand_convs.1.9.bias',
              'roi_head.offset_head.fcs.0.weight', 'roi_head.offset_head.fc_offset.weight']:
        g = gr[n]
        out['gradnorm_' + n] = np.float32(g.norm().item())
        out['gradhead_' + n] = T(g.reshape(-1)[:16])
    # every parameter: gradient norm + first 8 entries (the fp32 parity mode's backward is checked against ALL of them)
    for n, g in gr.items():
        out['allnorm_' + n] = np.float32(g.norm().item())
        out['allhead_' + n] = T(g.reshape(-1)[:8])
    out['meta'] = np.array([size, batch, num_gt])
    np.savez_compressed(os.path.join(GOLD, f'e2e_{size}.npz'), **out)
    print(f'e2e_{size}.npz', {k: float(v) for k, v in log_vars.items()})


def e2e_test(size=256):
    """Reference LOFT in eval mode: forward_test -> simple_test 3-tuple on a seeded tile."""
    from bonai_amd.config import Config
    from bonai_amd.synth import make_batch
    from mmdet.models import build_detector
    from oracle.synth_weights import synth_state_dict
    cfg = Config.fromfile('/root/reference/configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py')
    cfg.model.pretrained = None
    m = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    m.load_state_dict(synth_state_dict(m.state_dict()))
    m.eval()
    data = make_batch(1, size, 4)
    with torch.no_grad():
        bbox_results, segm_results, offset_results = m(img=[data['img']], img_metas=[data['img_metas']], return_loss=False,
                                                       rescale=True)
    det = bbox_results[0]
    masks = np.stack(segm_results[0]) if len(segm_results[0]) else np.zeros((0, size, size), bool)
    out = dict(det=det.astype(np.float32), offsets=np.asarray(offset_results, np.float32),
               mask_area=masks.reshape(masks.shape[0], -1).sum(1).astype(np.int64),
               mask_rowsum=masks.sum(2).astype(np.int32)[:64], meta=np.array([size]))
    np.savez_compressed(os.path.join(GOLD, f'e2e_test_{size}.npz'), **out)
    print(f'e2e_test_{size}.npz dets', det.shape, 'top', det[:2], 'offsets', out['offsets'][:2], 'areas', out['mask_area'][:5])


def hrnet(size=128):
    """Reference HRNetV2p-W32 backbone + HRFPN neck (mmdet/models/backbones/hrnet.py, necks/hrfpn.py) on a seeded tile with
    name-keyed synthetic weights: outputs (crops + sums) and a few parameter gradients of sum(neck outputs * ramp)."""
    from mmdet.models.backbones.hrnet import HRNet
    from mmdet.models.necks.hrfpn import HRFPN
    from bonai_amd.synth import make_batch
    from oracle.loft_model_ref import HRNET_W32
    from oracle.synth_weights import synth_tensor
    bb = HRNet(extra={k: dict(v) for k, v in HRNET_W32.items()})
    neck = HRFPN(in_channels=[32, 64, 128, 256], out_channels=256)
    bb.load_state_dict({k: synth_tensor('backbone.' + k, v.shape) for k, v in bb.state_dict().items()})
    neck.load_state_dict({k: synth_tensor('neck.' + k, v.shape) for k, v in neck.state_dict().items()})
    bb.train(); neck.train()                  # norm_eval=True keeps the BN statistics frozen (hrnet.py:527-537)
    img = make_batch(1, size, 4)['img']
    ys = bb(img)
    outs = neck(ys)
    out = dict(meta=np.array([size]), n_backbone_keys=np.array([len(bb.state_dict())]), n_neck_keys=np.array([len(neck.state_dict())]))
    for i, y in enumerate(ys):
        out[f'bb_{i}_shape'] = np.array(y.shape)
        out[f'bb_{i}_crop'] = T(y[:, :8, :6, :6])
        out[f'bb_{i}_absmean'] = T(y.abs().mean())
        out[f'bb_{i}_sum'] = T(y.double().sum())
    for i, o in enumerate(outs):
        out[f'neck_{i}_shape'] = np.array(o.shape)
        out[f'neck_{i}_crop'] = T(o[:, :8, :6, :6])
        out[f'neck_{i}_absmean'] = T(o.abs().mean())
        out[f'neck_{i}_sum'] = T(o.double().sum())
    loss = sum((o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum() for o in outs) / 1000.0
    loss.backward()
    out['loss'] = T(loss)
    names = ['backbone.conv1.weight', 'backbone.bn1.weight', 'backbone.conv2.weight', 'backbone.layer1.0.conv1.weight',
             'backbone.transition1.0.0.weight', 'backbone.stage2.0.branches.0.0.conv1.weight',
             'backbone.stage2.0.fuse_layers.0.1.0.weight', 'backbone.stage3.1.fuse_layers.2.0.1.0.weight',
             'backbone.stage3.0.branches.2.3.bn2.bias', 'backbone.stage4.2.branches.3.3.conv2.weight',
             'backbone.stage4.0.fuse_layers.0.3.1.weight', 'neck.reduction_conv.conv.weight', 'neck.fpn_convs.4.conv.bias']
    params = {('backbone.' + k): v for k, v in bb.named_parameters()}
    params.update({('neck.' + k): v for k, v in neck.named_parameters()})
    for n in names:
        out['gradnorm_' + n] = T(params[n].grad.norm())
        out['gradhead_' + n] = T(params[n].grad.reshape(-1)[:16])
    np.savez_compressed(os.path.join(GOLD, f'hrnet_{size}.npz'), **out)
    print(f'hrnet_{size}.npz', [tuple(y.shape) for y in ys], [float(out[f'neck_{i}_absmean']) for i in range(5)], float(loss))


def offset_head():
    """Reference OffsetHead (attribute_heads/offset_head.py:23-265, the basic LOFT head without FOA): forward + loss + backward,
    get_targets and get_offsets for the rectangular reg_num=2 head (SmoothL1 x16 as configs/loft_foa use, and the default MSE)
    and the polar reg_num=3 variant.  Inputs are regenerated from names (synth_tensor), only expected outputs are stored."""
    from mmdet.models.roi_heads.attribute_heads.offset_head import OffsetHead
    from oracle.synth_weights import synth_tensor
    out = {}
    rng = np.random.RandomState(7)

    class _Res:
        pass

    def boxes(n, size=1024.):
        cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
        w, h = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
        return torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size), dtype=torch.float32)
    res, gt_offs = [], []
    for i, n in enumerate((5, 0, 3)):                       # the middle image has no positives (offset_head.py:142-143)
        r = _Res()
        r.pos_bboxes = boxes(n)
        r.pos_assigned_gt_inds = torch.tensor(rng.randint(0, 4, n), dtype=torch.long)
        res.append(r)
        gt_offs.append(torch.tensor(rng.uniform(-40, 40, (4, 2)), dtype=torch.float32))
        out[f'pos_{i}'], out[f'ind_{i}'], out[f'gtoff_{i}'] = T(r.pos_bboxes), T(r.pos_assigned_gt_inds), T(gt_offs[-1])
    det = boxes(8)
    det[0] = torch.tensor([0., 0., 1024., 1024.])           # decode clamp at +-1024
    out['det'] = T(det)
    for tag, kw, nconv in (('rect', dict(reg_num=2, loss_offset=dict(type='SmoothL1Loss', loss_weight=16.0)), 2),
                           ('mse', dict(reg_num=2), 1),
                           ('polar', dict(reg_num=3, offset_coordinate='polar',
                                          loss_offset=dict(type='SmoothL1Loss', loss_weight=16.0)), 1)):
        head = OffsetHead(num_convs=nconv, **kw)
        sd = {k: synth_tensor('roi_head.offset_head.' + k, v.shape) for k, v in head.state_dict().items()}
        sd['fc_offset.weight'] = sd['fc_offset.weight'] * 30.0   # O(1) predictions so the SmoothL1 knee and the clamp are hit
        head.load_state_dict(sd)
        x = (synth_tensor(f'offset_head.x.{tag}', (8, 256, 7, 7)) * 1.0).requires_grad_(True)
        pred = head(x)
        tg = head.get_targets(res, gt_offs, None)
        loss = head.loss(pred, tg)['loss_offset']
        loss.backward()
        out[f'{tag}_pred'], out[f'{tag}_targets'], out[f'{tag}_loss'] = T(pred), T(tg), T(loss)
        out[f'{tag}_offsets'] = head.get_offsets(pred.detach(), det, None, False)
        out[f'{tag}_empty_shape'] = np.array(head(x[:0]).shape)
        for n, p_ in head.named_parameters():
            if n in ('convs.0.weight', 'convs.0.bias', 'fcs.0.weight', 'fcs.1.bias', 'fc_offset.weight', 'fc_offset.bias'):
                out[f'{tag}_gradnorm_{n}'] = T(p_.grad.norm())
                out[f'{tag}_gradhead_{n}'] = T(p_.grad.reshape(-1)[:16])
        out[f'{tag}_gradx_crop'] = T(x.grad[:, :8, :3, :3])
        out[f'{tag}_gradx_norm'] = T(x.grad.norm())
    np.savez_compressed(os.path.join(GOLD, 'offset_head.npz'), **out)
    print('offset_head.npz', len(out), 'arrays', {k: float(out[k]) for k in out if k.endswith('_loss')}, out['polar_offsets'][:2])


def foa_head():
    """Reference OffsetHeadExpandFeature (attribute_heads/offset_head_expand_feature.py:25-344) stand-alone: forward, get_targets,
    loss and backward with the FC stack shared by the four rotation branches (configs/loft_foa) and with one stack per branch
    (``share_expand_fc=False``, the class default; :82-95, :147-152).  Inputs are regenerated from names (synth_tensor)."""
    from mmdet.models.roi_heads.attribute_heads.offset_head_expand_feature import OffsetHeadExpandFeature
    from oracle.synth_weights import synth_tensor
    out = {}
    rng = np.random.RandomState(11)

    class _Res:
        pass

    def boxes(n, size=1024.):
        cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
        w, h = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
        return torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size), dtype=torch.float32)
    res, gt_offs = [], []
    for i, n in enumerate((4, 0, 2)):                       # the middle image has no positives
        r = _Res()
        r.pos_bboxes = boxes(n)
        r.pos_assigned_gt_inds = torch.tensor(rng.randint(0, 3, n), dtype=torch.long)
        res.append(r)
        gt_offs.append(torch.tensor(rng.uniform(-40, 40, (3, 2)), dtype=torch.float32))
        out[f'pos_{i}'], out[f'ind_{i}'], out[f'gtoff_{i}'] = T(r.pos_bboxes), T(r.pos_assigned_gt_inds), T(gt_offs[-1])
    for tag, kw, nconv in (('unshared', dict(share_expand_fc=False, loss_offset=dict(type='SmoothL1Loss', loss_weight=16.0)), 1),
                           ('shared', dict(share_expand_fc=True), 2)):
        head = OffsetHeadExpandFeature(num_convs=nconv, **kw)
        sd = {k: synth_tensor('roi_head.offset_head.' + k, v.shape) for k, v in head.state_dict().items()}
        for k in sd:
            if 'fc_offset' in k and k.endswith('weight'):
                sd[k] = sd[k] * 30.0                        # O(1) predictions so the SmoothL1 knee is crossed
        head.load_state_dict(sd)
        out[f'{tag}_state_keys'] = np.array(sorted(sd))
        x = synth_tensor(f'foa_head.x.{tag}', (6, 256, 7, 7)).requires_grad_(True)
        pred = head(x)
        tg = head.get_targets(res, gt_offs, None)
        loss = head.loss(pred, tg)['loss_offset']
        loss.backward()
        out[f'{tag}_pred'], out[f'{tag}_targets'], out[f'{tag}_loss'] = T(pred), T(tg), T(loss)
        out[f'{tag}_empty_shape'] = np.array(head(x[:0]).shape)
        keep = ('expand_convs.0.0.weight', 'expand_convs.3.0.bias', 'expand_fcs.1.0.weight', 'expand_fcs.2.1.bias',
                'expand_fc_offsets.3.weight', 'expand_fc_offsets.0.bias', 'fcs.0.weight', 'fcs.1.bias', 'fc_offset.weight')
        for n, p_ in head.named_parameters():
            if n in keep:
                out[f'{tag}_gradnorm_{n}'] = T(p_.grad.norm())
                out[f'{tag}_gradhead_{n}'] = T(p_.grad.reshape(-1)[:16])
        out[f'{tag}_gradx_crop'] = T(x.grad[:, :8, :3, :3])
        out[f'{tag}_gradx_norm'] = T(x.grad.norm())
    np.savez_compressed(os.path.join(GOLD, 'foa_head.npz'), **out)
    print('foa_head.npz', len(out), 'arrays', {k: float(out[k]) for k in out if k.endswith('_loss')})


def foa_variants():
    """Reference OffsetHeadExpandFeature with the rotation sets its offset_fusion spells out besides the four-branch one
    (offset_head_expand_feature.py:371-385: [0, 180], [0, 90], [0, 90, 180]) -- forward, get_targets, loss, backward, offset_fusion
    in both models ('max' :370-397, 'mean' :358-369) and get_offsets -- and the 'mean' fusion of the four-branch head.  Inputs are
    regenerated from names (synth_tensor)."""
    from mmdet.models.roi_heads.attribute_heads.offset_head_expand_feature import OffsetHeadExpandFeature
    from oracle.synth_weights import synth_tensor
    out = {}
    rng = np.random.RandomState(23)

    class _Res:
        pass

    def boxes(n, size=1024.):
        cx, cy = rng.uniform(0, size, n), rng.uniform(0, size, n)
        w, h = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
        return torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).clip(0, size), dtype=torch.float32)
    res, gt_offs = [], []
    for i, n in enumerate((3, 0, 2)):
        r = _Res()
        r.pos_bboxes = boxes(n)
        r.pos_assigned_gt_inds = torch.tensor(rng.randint(0, 3, n), dtype=torch.long)
        res.append(r)
        gt_offs.append(torch.tensor(rng.uniform(-40, 40, (3, 2)), dtype=torch.float32))
        out[f'pos_{i}'], out[f'ind_{i}'], out[f'gtoff_{i}'] = T(r.pos_bboxes), T(r.pos_assigned_gt_inds), T(gt_offs[-1])
    det = boxes(5)
    out['det_bboxes'] = T(det)
    for tag, rots in (('r0_180', [0, 180]), ('r0_90', [0, 90]), ('r0_90_180', [0, 90, 180]), ('r4', [0, 90, 180, 270])):
        head = OffsetHeadExpandFeature(num_convs=1, share_expand_fc=True, expand_feature_num=len(rots), rotations=list(rots))
        sd = {k: synth_tensor('roi_head.offset_head.' + k, v.shape) for k, v in head.state_dict().items()}
        for k in sd:
            if 'fc_offset' in k and k.endswith('weight'):
                sd[k] = sd[k] * 30.0
        head.load_state_dict(sd)
        out[f'{tag}_state_keys'] = np.array(sorted(sd))
        x = synth_tensor(f'foa_variants.x.{tag}', (5, 256, 7, 7)).requires_grad_(True)
        pred = head(x)
        tg = head.get_targets(res, gt_offs, None)
        loss = head.loss(pred, tg)['loss_offset']
        loss.backward()
        out[f'{tag}_pred'], out[f'{tag}_targets'], out[f'{tag}_loss'] = T(pred), T(tg), T(loss)
        out[f'{tag}_empty_shape'] = np.array(head(x[:0]).shape)
        for n, p_ in head.named_parameters():
            if n in ('expand_convs.0.0.weight', f'expand_convs.{len(rots) - 1}.0.bias', 'fcs.0.weight', 'fc_offset.weight'):
                out[f'{tag}_gradnorm_{n}'] = T(p_.grad.norm())
        out[f'{tag}_gradx_norm'] = T(x.grad.norm())
        with torch.no_grad():
            p = pred.detach().clone()
            p[0, 0] = 0.0                                    # an exact zero in the main branch: polarity -1 (:405-407)
            out[f'{tag}_fuse_in'] = T(p)
            out[f'{tag}_fuse_max'] = T(head.offset_fusion(p, model='max'))
            out[f'{tag}_fuse_mean'] = T(head.offset_fusion(p, model='mean'))
            out[f'{tag}_get_offsets'] = np.asarray(head.get_offsets(p, det, None, False, img_shape=[1024, 1024]))
    np.savez_compressed(os.path.join(GOLD, 'foa_variants.npz'), **out)
    print('foa_variants.npz', len(out), 'arrays', {k: float(out[k]) for k in out if k.endswith('_loss')})


def data_pipeline():
    """Reference BONAI._parse_ann_info (bonai.py:105-256) and RandomFlip.bbox_flip/offset_flip (transforms.py:379-466) on
    synthetic annotations; the expected outputs are stored, the inputs are regenerated by synth_bonai_anns()."""
    from mmdet.datasets.bonai import BONAI
    from mmdet.datasets.pipelines.transforms import RandomFlip
    from bonai_amd.synth import synth_bonai_anns
    out = {}
    img_info = dict(width=1024, height=1024, filename='L18_104400_210392.png')
    for tag, kw in (('roof', dict(bbox_type='roof', mask_type='roof', offset_coordinate='rectangle')),
                    ('building_polar', dict(bbox_type='building', mask_type='footprint', offset_coordinate='polar')),
                    ('footprint', dict(bbox_type='footprint', mask_type='roof', offset_coordinate='rectangle'))):
        ds = BONAI.__new__(BONAI)
        ds.cat_ids, ds.cat2label, ds.resolution, ds.ignore_buildings = [1], {1: 0}, 0.6, True
        for k, v in kw.items():
            setattr(ds, k, v)
        ann = ds._parse_ann_info(img_info, synth_bonai_anns())
        for k in ('bboxes', 'labels', 'bboxes_ignore', 'offsets', 'building_heights', 'roof_bboxes', 'footprint_bboxes'):
            out[f'{tag}_{k}'] = np.asarray(ann[k])
        out[f'{tag}_angle'] = np.array(ann['angle'], dtype=np.float64)
        out[f'{tag}_only_footprint_flag'] = np.array(ann['only_footprint_flag'], dtype=np.float64)
        out[f'{tag}_n_masks'] = np.array([len(ann['masks']), len(ann['roof_masks']), len(ann['footprint_masks'])])
        out[f'{tag}_mask0'] = np.asarray(ann['masks'][0], dtype=np.float64).reshape(-1)
        out[f'{tag}_mask_of_only_fp'] = np.asarray(ann['masks'][-3], dtype=np.float64).reshape(-1)
    empty = ds._parse_ann_info(img_info, [])
    out['empty_heights_shape'] = np.array(empty['building_heights'].shape)
    out['empty_angle'] = np.array(empty['angle'])
    rf = RandomFlip(flip_ratio=0.5)
    bb = out['roof_bboxes']
    off = out['roof_offsets']
    for d in ('horizontal', 'vertical'):
        out[f'flip_{d}_bboxes'] = rf.bbox_flip(bb, (1024, 1024, 3), d)
        out[f'flip_{d}_offsets'] = rf.offset_flip(off, (1024, 1024, 3), d)
    np.savez_compressed(os.path.join(GOLD, 'data_pipeline.npz'), **out)
    print('data_pipeline.npz', out['roof_bboxes'].shape, out['building_polar_offsets'][:2], float(out['roof_angle']))


if __name__ == '__main__':
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'data':
        data_pipeline()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'hrnet':
        hrnet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'e2e':
        e2e()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'offset_head':
        offset_head()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'foa_head':
        foa_head()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'foa_variants':
        foa_variants()
        sys.exit(0)
    core_ops()
    e2e()
    e2e_test()
    hrnet()
    offset_head()
    foa_head()
    foa_variants()
    data_pipeline()
