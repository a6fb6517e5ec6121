"""End-to-end CPU restatement (torch-CPU fp32) of the reference's LOFT R50-FPN training forward.
TEST INFRASTRUCTURE: the parity checker for the HIP path and bench.py's cpu_baseline ("port").

Functional style over a plain ``state_dict`` with the reference's key names; per-image python loops
exactly where the reference has them.  Follows (paths relative to /root/reference/mmdet):
  models/detectors/two_stage.py:105-167 (forward_train), base.py:175-208 (_parse_losses)
  models/backbones/resnet.py:266-298,623-638; models/necks/fpn.py:170-199
  models/dense_heads/rpn_head.py:38-44,79-168; anchor_head.py:180-497
  models/roi_heads/loft_roi_head.py:44-194; standard_roi_head.py:148-161
  models/roi_heads/bbox_heads/convfc_bbox_head.py:135-173, bbox_head.py:84-185
  models/roi_heads/mask_heads/fcn_mask_head.py:118-149; core/mask/mask_target.py:33-62
  models/roi_heads/attribute_heads/offset_head_expand_feature.py:134-205,271-344
Pinned against the reference itself by tests/golden/e2e_256.npz (oracle/ref_harness/make_goldens.py).
The mmcv-1.0.5 ops (RoIAlign, nms) are the plain-C restatement in both (parity unpinned there).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import cops
from . import ops_ref as R

STRIDES = (4, 8, 16, 32, 64)
STAGE_BLOCKS = (3, 4, 6, 3)


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], False, 0., eps)


# ---- numerics mode -------------------------------------------------------------------------------------------------------
# None (default): the reference's fp32 arithmetic, operation for operation (what tests/golden/e2e_256.npz pins).
# torch.bfloat16 / torch.float16 ("16-bit points" mode, R50-FPN path): the SAME graph with values rounded to the 16-bit type at
# exactly the points where the HIP training path (bonai_amd/nn.py, csrc/*.hip) holds 16-bit data:
#   * every conv / deconv / linear operand: activations as stored (16-bit), weights AFTER folding the frozen-statistics BN
#     (w * gamma / sqrt(var + eps), loft_fold_pack) rounded to 16 bit; accumulation, bias (= BN shift), residual add and ReLU
#     in fp32; the result rounded to 16 bit when the layer's output tensor is 16-bit (all but the narrow heads: rpn_cls /
#     rpn_reg, fc_cls / fc_reg, conv_logits, fc_offset, which stay fp32 like the reference's force_fp32 boundary);
#   * the shortcut conv of a downsample block is rounded before it enters conv3's epilogue as the residual (_ResBlockFn);
#   * the image enters the stem rounded to 16 bit (stem_mfma_kernel builds its im2col rows in the 16-bit type);
#   * FPN top-down: each in-place sum is rounded (loft_upsample2x_add on 16-bit maps);
#   * RoIAlign: fp32 arithmetic on the 16-bit maps, 16-bit output.
# Losses, coders, assignment, NMS stay fp32.  The HIP kernels accumulate in another ORDER, so an output can land on the other
# side of a rounding boundary (1 ulp = 2^-8 relative for bf16): agreement is statistical -- relative L2 error of a tensor at a
# few 1e-3 -- not element-exact.  tests/golden/e2e_256_bf16.npz (oracle/make_bf16_golden.py) holds this mode's outputs.
_NUM = [None]


class numerics:
    """``with numerics(torch.bfloat16): forward_train(...)``"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev, _NUM[0] = _NUM[0], self.dtype

    def __exit__(self, *a):
        _NUM[0] = self.prev


def _q(x):
    return x if _NUM[0] is None else x.to(_NUM[0]).float()


def _layer(x, w, b=None, stride=1, pad=0, bn=None, relu=False, res=None, out16=True, transposed=False):
    """conv (+ BN) (+ residual) (+ ReLU).  fp32 mode: the reference's op order.  16-bit mode: see above.  bn = (sd, prefix)."""
    conv = (lambda a, ww, bb: F.conv_transpose2d(a, ww, bb, stride=stride)) if transposed else \
        (lambda a, ww, bb: F.conv2d(a, ww, bb, stride, pad))
    if _NUM[0] is None:
        y = conv(x, w, b)
        if bn is not None:
            y = _bn(y, bn[0], bn[1])
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y
    shift = b
    if bn is not None:
        sd, p = bn
        scale = sd[p + '.weight'] / torch.sqrt(sd[p + '.running_var'] + 1e-5)
        w = w * scale[:, None, None, None]
        shift = sd[p + '.bias'] - sd[p + '.running_mean'] * scale + (0 if b is None else b * scale)
    y = conv(_q(x), _q(w), None)
    if shift is not None:
        y = y + shift[None, :, None, None]
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return _q(y) if out16 else y


def _lin(x, w, b, relu=False, out16=True):
    if _NUM[0] is None:
        y = F.linear(x, w, b)
        return F.relu(y) if relu else y
    y = F.linear(_q(x), _q(w), None) + b
    if relu:
        y = F.relu(y)
    return _q(y) if out16 else y


def stem(sd, img, prefix='backbone.'):
    """conv1 7x7/2 + bn1 + relu + maxpool 3x3/2 (resnet.py:623-630)."""
    x = _layer(img, sd[prefix + 'conv1.weight'], None, 2, 3, bn=(sd, prefix + 'bn1'), relu=True)
    return F.max_pool2d(x, 3, 2, 1)


def res_stage(sd, x, li, prefix='backbone.'):
    """layer{li+1}: STAGE_BLOCKS[li] bottlenecks (resnet.py:266-298)."""
    for bi in range(STAGE_BLOCKS[li]):
        p = f'{prefix}layer{li + 1}.{bi}.'
        stride = 2 if (bi == 0 and li > 0) else 1
        out = _layer(x, sd[p + 'conv1.weight'], bn=(sd, p + 'bn1'), relu=True)
        if (p + 'conv2.conv_offset.weight') in sd:      # DCNv2 (resnet.py:171-194)
            if _NUM[0] is not None:
                raise NotImplementedError('16-bit points mode: plain R50-FPN only')
            c2 = R.mdcn_pack(out, sd[p + 'conv2.weight'], None, sd[p + 'conv2.conv_offset.weight'],
                             sd[p + 'conv2.conv_offset.bias'], stride, 1)
            out = F.relu(_bn(c2, sd, p + 'bn2'))
        else:
            out = _layer(out, sd[p + 'conv2.weight'], None, stride, 1, bn=(sd, p + 'bn2'), relu=True)
        idt = x
        if (p + 'downsample.0.weight') in sd:
            idt = _layer(x, sd[p + 'downsample.0.weight'], None, stride, bn=(sd, p + 'downsample.1'))
        x = _layer(out, sd[p + 'conv3.weight'], bn=(sd, p + 'bn3'), res=idt, relu=True)
    return x


def backbone(sd, img, prefix='backbone.'):
    x = stem(sd, img, prefix)
    outs = []
    for li in range(len(STAGE_BLOCKS)):
        x = res_stage(sd, x, li, prefix)
        outs.append(x)
    return outs


# ---- HRNet-W32 + HRFPN (BASELINE config 5): mmdet/models/backbones/hrnet.py:12-537, mmdet/models/necks/hrfpn.py:11-102
HRNET_W32 = dict(stage1=dict(num_modules=1, num_branches=1, block='BOTTLENECK', num_blocks=(4,), num_channels=(64,)),
                 stage2=dict(num_modules=1, num_branches=2, block='BASIC', num_blocks=(4, 4), num_channels=(32, 64)),
                 stage3=dict(num_modules=4, num_branches=3, block='BASIC', num_blocks=(4, 4, 4), num_channels=(32, 64, 128)),
                 stage4=dict(num_modules=3, num_branches=4, block='BASIC', num_blocks=(4, 4, 4, 4),
                             num_channels=(32, 64, 128, 256)))


def _cb(sd, p, x, stride=1, pad=0, relu=False):
    """Sequential(conv '0', bn '1'[, relu]) (hrnet.py:130-172, 345-391)."""
    y = _bn(F.conv2d(x, sd[p + '0.weight'], None, stride, pad), sd, p + '1')
    return F.relu(y) if relu else y


def _basic_block(sd, p, x):
    """resnet.py:65-92 (BasicBlock.forward), stride 1."""
    out = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight'], None, 1, 1), sd, p + 'bn1'))
    out = _bn(F.conv2d(out, sd[p + 'conv2.weight'], None, 1, 1), sd, p + 'bn2')
    idt = _cb(sd, p + 'downsample.', x) if (p + 'downsample.0.weight') in sd else x
    return F.relu(out + idt)


def _bottleneck(sd, p, x):
    out = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight']), sd, p + 'bn1'))
    out = F.relu(_bn(F.conv2d(out, sd[p + 'conv2.weight'], None, 1, 1), sd, p + 'bn2'))
    out = _bn(F.conv2d(out, sd[p + 'conv3.weight']), sd, p + 'bn3')
    idt = _cb(sd, p + 'downsample.', x) if (p + 'downsample.0.weight') in sd else x
    return F.relu(out + idt)


def _hr_module(sd, p, xs, num_blocks):
    """HRModule.forward (hrnet.py:177-195)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(num_blocks[i]):
            xs[i] = _basic_block(sd, f'{p}branches.{i}.{k}.', xs[i])
    if nb == 1:
        return xs
    outs = []
    for i in range(nb):
        y = 0
        for j in range(nb):
            if i == j:
                y = y + xs[j]
            elif j > i:
                t = _cb(sd, f'{p}fuse_layers.{i}.{j}.', xs[j])
                y = y + F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _cb(sd, f'{p}fuse_layers.{i}.{j}.{k}.', t, 2, 1, relu=(k != i - j - 1))
                y = y + t
        outs.append(F.relu(y))
    return outs


def hrnet_backbone(sd, img, extra=HRNET_W32, prefix='backbone.'):
    """HRNet.forward (hrnet.py:480-515)."""
    if _NUM[0] is not None:
        raise NotImplementedError('16-bit points mode: plain R50-FPN only')
    x = F.relu(_bn(F.conv2d(img, sd[prefix + 'conv1.weight'], None, 2, 1), sd, prefix + 'bn1'))
    x = F.relu(_bn(F.conv2d(x, sd[prefix + 'conv2.weight'], None, 2, 1), sd, prefix + 'bn2'))
    for k in range(extra['stage1']['num_blocks'][0]):
        x = _bottleneck(sd, f'{prefix}layer1.{k}.', x)
    y = [x]
    for si in (2, 3, 4):
        cfg = extra[f'stage{si}']
        xs = []
        for i in range(cfg['num_branches']):
            tp = f'{prefix}transition{si - 1}.{i}.'
            if (tp + '0.weight') in sd:                    # existing branch, channel change: Sequential(conv, bn, relu)
                xs.append(_cb(sd, tp, y[-1], 1, 1, relu=True))
            elif (tp + '0.0.weight') in sd:                # new branch: chain of stride-2 conv-bn-relu
                t, j = y[-1], 0
                while (f'{tp}{j}.0.weight') in sd:
                    t = _cb(sd, f'{tp}{j}.', t, 2, 1, relu=True)
                    j += 1
                xs.append(t)
            else:
                xs.append(y[i])
        for m in range(cfg['num_modules']):
            xs = _hr_module(sd, f'{prefix}stage{si}.{m}.', xs, cfg['num_blocks'])
        y = xs
    return y


def hrfpn(sd, feats, num_outs=5, prefix='neck.'):
    """HRFPN.forward (hrfpn.py:78-102)."""
    outs = [feats[0]] + [F.interpolate(f, scale_factor=2 ** i, mode='bilinear') for i, f in enumerate(feats) if i > 0]
    out = F.conv2d(torch.cat(outs, 1), sd[prefix + 'reduction_conv.conv.weight'], sd[prefix + 'reduction_conv.conv.bias'])
    pyr = [out] + [F.avg_pool2d(out, 2 ** i, 2 ** i) for i in range(1, num_outs)]
    return [F.conv2d(o, sd[f'{prefix}fpn_convs.{i}.conv.weight'], sd[f'{prefix}fpn_convs.{i}.conv.bias'], padding=1)
            for i, o in enumerate(pyr)]


def extract_feat(sd, img):
    """backbone + neck by the checkpoint's key set (ResNet-50 + FPN, or HRNet + HRFPN)."""
    if 'backbone.stage2.0.branches.0.0.conv1.weight' in sd:
        return hrfpn(sd, hrnet_backbone(sd, img))
    return fpn(sd, backbone(sd, img))


def _fpn_conv(sd, p, x, pad):
    """ConvModule of the neck: plain conv, or DCNv2 when conv_cfg=dict(type='DCNv2') (fpn.py:116-132)."""
    if (p + 'conv_offset.weight') in sd:
        return R.mdcn_pack(x, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'conv_offset.weight'], sd[p + 'conv_offset.bias'], 1, pad)
    return _layer(x, sd[p + 'weight'], sd[p + 'bias'], 1, pad)


def fpn(sd, feats, prefix='neck.'):
    lats = [_fpn_conv(sd, f'{prefix}lateral_convs.{i}.conv.', f, 0) for i, f in enumerate(feats)]
    for i in range(len(lats) - 1, 0, -1):
        lats[i - 1] = _q(lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest'))
    outs = [_fpn_conv(sd, f'{prefix}fpn_convs.{i}.conv.', l, 1) for i, l in enumerate(lats)]
    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return outs


def rpn_forward(sd, feats, prefix='rpn_head.'):
    cls, reg = [], []
    for x in feats:
        h = _layer(x, sd[prefix + 'rpn_conv.weight'], sd[prefix + 'rpn_conv.bias'], 1, 1, relu=True)
        cls.append(_layer(h, sd[prefix + 'rpn_cls.weight'], sd[prefix + 'rpn_cls.bias'], out16=False))
        reg.append(_layer(h, sd[prefix + 'rpn_reg.weight'], sd[prefix + 'rpn_reg.bias'], out16=False))
    return cls, reg


def rpn_loss(cls, reg, gt_bboxes, choose=R.choose_first):
    B = cls[0].shape[0]
    sizes = [tuple(c.shape[2:]) for c in cls]
    anchors = torch.cat(R.grid_anchors(sizes, STRIDES), 0)
    flat_cls = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1) for c in cls], 1)
    flat_reg = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in reg], 1)
    labels, lw, bt, bw, npos, nneg = [], [], [], [], 0, 0
    for i in range(B):
        gi, _ = R.max_iou_assign(anchors, gt_bboxes[i], 0.7, 0.3, 0.3, True)
        pos, neg = R.sample(gi, 512, 0.5, choose)
        lab = torch.zeros(anchors.shape[0], dtype=torch.long)
        w = torch.zeros(anchors.shape[0])
        t = torch.zeros(anchors.shape[0], 4)
        tw = torch.zeros(anchors.shape[0], 4)
        if pos.numel():
            t[pos] = R.bbox2delta(anchors[pos], gt_bboxes[i][gi[pos] - 1])
            tw[pos] = 1.0
            lab[pos] = 1
            w[pos] = 1.0
        if neg.numel():
            w[neg] = 1.0
        labels.append(lab); lw.append(w); bt.append(t); bw.append(tw)
        npos += max(pos.numel(), 1); nneg += max(neg.numel(), 1)
    avg = float(npos + nneg)
    labels, lw, bt, bw = torch.stack(labels), torch.stack(lw), torch.stack(bt), torch.stack(bw)
    loss_cls = R.bce_sigmoid_loss(flat_cls.reshape(-1, 1), labels.reshape(-1), lw.reshape(-1), avg)
    loss_bbox = R.l1_loss(flat_reg.reshape(-1, 4), bt.reshape(-1, 4), bw.reshape(-1, 4), avg)
    return loss_cls, loss_bbox


def rpn_proposals(cls, reg, img_shape, nms_pre=3000, nms_post=3000, nms_thr=0.7):
    """rpn_head.py:79-168, per image."""
    B = cls[0].shape[0]
    sizes = [tuple(c.shape[2:]) for c in cls]
    mlvl_anchors = R.grid_anchors(sizes, STRIDES)
    out = []
    for i in range(B):
        scores, deltas, ancs, ids = [], [], [], []
        for l in range(len(cls)):
            s = cls[l][i].detach().permute(1, 2, 0).reshape(-1).sigmoid()
            d = reg[l][i].detach().permute(1, 2, 0).reshape(-1, 4)
            a = mlvl_anchors[l]
            if nms_pre > 0 and s.shape[0] > nms_pre:
                order = cops.argsort_desc(s)[:nms_pre]
                s, d, a = s[order], d[order], a[order]
            scores.append(s); deltas.append(d); ancs.append(a)
            ids.append(torch.full((s.shape[0],), l, dtype=torch.long))
        scores, deltas, ancs, ids = torch.cat(scores), torch.cat(deltas), torch.cat(ancs), torch.cat(ids)
        props = R.delta2bbox(ancs, deltas, max_shape=img_shape)
        dets, _ = cops.batched_nms(props, scores, ids, dict(type='nms', iou_threshold=nms_thr))
        out.append(dets[:nms_post])
    return out


def _fc(sd, p, x, relu=False, out16=True):
    """nn.Linear (+ ReLU).  out16=False: a head output (fp32 in every mode)."""
    return _lin(x, sd[p + '.weight'], sd[p + '.bias'], relu, out16)


def bbox_head(sd, x, prefix='roi_head.bbox_head.'):
    h = x.flatten(1)
    h = _fc(sd, prefix + 'shared_fcs.0', h, relu=True)
    h = _fc(sd, prefix + 'shared_fcs.1', h, relu=True)
    return _fc(sd, prefix + 'fc_cls', h, out16=False), _fc(sd, prefix + 'fc_reg', h, out16=False)


def mask_head(sd, x, prefix='roi_head.mask_head.'):
    for i in range(4):
        x = _layer(x, sd[f'{prefix}convs.{i}.conv.weight'], sd[f'{prefix}convs.{i}.conv.bias'], 1, 1, relu=True)
    x = _layer(x, sd[prefix + 'upsample.weight'], sd[prefix + 'upsample.bias'], 2, 0, relu=True, transposed=True)
    return _layer(x, sd[prefix + 'conv_logits.weight'], sd[prefix + 'conv_logits.bias'], out16=False)


def foa_head(sd, x, num_convs=10, prefix='roi_head.offset_head.', share_expand_fc=True, num_fcs=2, rotations=(0, 90, 180, 270)):
    """offset_head_expand_feature.py:134-161: per rotation branch num_convs x (conv3x3 + ReLU), flatten, num_fcs x (FC + ReLU),
    FC -> 2; the FC stack is `fcs` / `fc_offset` for every branch (share_expand_fc=True, configs/loft_foa) or the branch's own
    `expand_fcs.k.*` / `expand_fc_offsets.k` (:147-152).  Empty input -> (0, 2 * branches) (:135-136).  Branch k sees the feature
    rotated by rotations[k] degrees (:163-193)."""
    if x.shape[0] == 0:
        return x.new_empty(0, 2 * len(rotations))
    outs = []
    for k, rot in enumerate(rotations):
        h = R.foa_rotate_feature(x, rot // 90)
        for i in range(num_convs):
            h = _layer(h, sd[f'{prefix}expand_convs.{k}.{i}.weight'], sd[f'{prefix}expand_convs.{k}.{i}.bias'], 1, 1, relu=True)
        h = h.reshape(h.shape[0], -1)
        for i in range(num_fcs):
            h = _fc(sd, prefix + (f'fcs.{i}' if share_expand_fc else f'expand_fcs.{k}.{i}'), h, relu=True)
        outs.append(_fc(sd, prefix + ('fc_offset' if share_expand_fc else f'expand_fc_offsets.{k}'), h, out16=False))
    return torch.cat(outs, 0)


def offset_head(sd, x, num_convs=4, num_fcs=2, prefix='roi_head.offset_head.'):
    """roi_heads/attribute_heads/offset_head.py:90-106: num_convs x (conv3x3 + ReLU), flatten, num_fcs x (FC + ReLU), FC -> reg_num."""
    if x.shape[0] == 0:
        return x.new_empty(0, 2)
    for i in range(num_convs):
        x = _layer(x, sd[f'{prefix}convs.{i}.weight'], sd[f'{prefix}convs.{i}.bias'], 1, 1, relu=True)
    h = x.reshape(x.shape[0], -1)
    for i in range(num_fcs):
        h = _fc(sd, f'{prefix}fcs.{i}', h, relu=True)
    return _fc(sd, prefix + 'fc_offset', h, out16=False)


def roi_forward_train(sd, feats, proposals, gt_bboxes, gt_labels, gt_masks, gt_offsets, choose=R.choose_first,
                      num_classes=1):
    B = len(proposals)
    res = []
    for i in range(B):
        props = proposals[i][:, :4]
        gi, _ = R.max_iou_assign(props, gt_bboxes[i], 0.5, 0.5, 0.5, True)
        K = gt_bboxes[i].shape[0]
        boxes = torch.cat([gt_bboxes[i], props], 0)
        gi = torch.cat([torch.arange(1, K + 1, dtype=torch.long), gi])
        pos, neg = R.sample(gi, 1024, 0.25, choose)
        res.append(dict(pos_bboxes=boxes[pos], neg_bboxes=boxes[neg], pos_gt_inds=gi[pos] - 1,
                        pos_gt_bboxes=gt_bboxes[i][gi[pos] - 1], pos_gt_labels=gt_labels[i][gi[pos] - 1]))
    rois = R.bbox2roi([torch.cat([r['pos_bboxes'], r['neg_bboxes']]) for r in res])
    p4 = feats[:4]
    cls_score, bbox_pred = bbox_head(sd, _q(R.roi_extract(p4, rois, 7)))
    labels, lw, bt, bw = [], [], [], []
    for r in res:
        npos, nneg = r['pos_bboxes'].shape[0], r['neg_bboxes'].shape[0]
        lab = torch.full((npos + nneg,), num_classes, dtype=torch.long)
        lab[:npos] = r['pos_gt_labels']
        t = torch.zeros(npos + nneg, 4); w = torch.zeros(npos + nneg, 4)
        if npos:
            t[:npos] = R.bbox2delta(r['pos_bboxes'], r['pos_gt_bboxes'], stds=(.1, .1, .2, .2))
            w[:npos] = 1
        labels.append(lab); lw.append(torch.ones(npos + nneg)); bt.append(t); bw.append(w)
    labels, lw, bt, bw = torch.cat(labels), torch.cat(lw), torch.cat(bt), torch.cat(bw)
    losses = OrderedDict()
    losses['loss_cls'] = R.ce_loss(cls_score, labels, lw, max(float((lw > 0).sum()), 1.))
    losses['acc'] = R.accuracy(cls_score, labels)
    posm = labels < num_classes
    pred = bbox_pred.view(bbox_pred.shape[0], -1, 4)[posm, labels[posm]]
    losses['loss_bbox'] = R.l1_loss(pred, bt[posm], bw[posm], float(bt.shape[0])) if posm.any() else bbox_pred.sum() * 0
    pos_rois = R.bbox2roi([r['pos_bboxes'] for r in res])
    mask_pred = mask_head(sd, _q(R.roi_extract(p4, pos_rois, 14)))
    mts = []
    for i, r in enumerate(res):
        if r['pos_bboxes'].shape[0] == 0:
            mts.append(torch.zeros(0, 28, 28)); continue
        H, W = gt_masks[i].shape[1:]
        pb = r['pos_bboxes'].clone()
        pb[:, [0, 2]] = pb[:, [0, 2]].clamp(0, W); pb[:, [1, 3]] = pb[:, [1, 3]].clamp(0, H)
        rr = torch.cat([torch.arange(pb.shape[0], dtype=torch.float32)[:, None], pb], 1)
        sel = gt_masks[i][r['pos_gt_inds']].float()[:, None]
        mts.append((cops.roi_align_fwd(sel, rr, 28, 1.0, 0, True).squeeze(1) >= 0.5).float())
    mask_targets = torch.cat(mts)
    pos_labels = torch.cat([r['pos_gt_labels'] for r in res])
    losses['loss_mask'] = R.mask_bce_loss(mask_pred, mask_targets, pos_labels) if mask_pred.shape[0] else mask_pred.sum() * 0
    offset_pred = foa_head(sd, _q(R.roi_extract(p4, pos_rois, 7)))
    offset_targets = R.foa_offset_targets([r['pos_bboxes'] for r in res], [r['pos_gt_inds'] for r in res], gt_offsets)
    losses['loss_offset'] = 16.0 * R.smooth_l1_loss(offset_pred, offset_targets) if offset_pred.shape[0] else offset_pred.sum() * 0
    extras = dict(rois=rois, pos_rois=pos_rois, cls_score=cls_score, bbox_pred=bbox_pred, mask_pred=mask_pred,
                  mask_targets=mask_targets, offset_pred=offset_pred, offset_targets=offset_targets, labels=labels)
    return losses, extras


def forward_train(sd, img, gt_bboxes, gt_labels, gt_masks, gt_offsets, choose=R.choose_first, return_extras=False):
    """-> OrderedDict of the 7 losses + acc (+ 'loss'), keys as base.py:_parse_losses logs them."""
    H, W = img.shape[2:]
    feats = extract_feat(sd, img)
    cls, reg = rpn_forward(sd, feats)
    l_cls, l_box = rpn_loss(cls, reg, gt_bboxes, choose)
    props = rpn_proposals(cls, reg, (H, W, 3))
    losses = OrderedDict(loss_rpn_cls=l_cls, loss_rpn_bbox=l_box)
    roi_losses, extras = roi_forward_train(sd, feats, props, gt_bboxes, gt_labels, gt_masks, gt_offsets, choose)
    losses.update(roi_losses)
    losses['loss'] = sum(v.sum() for k, v in losses.items() if 'loss' in k)
    if return_extras:
        extras.update(feats=feats, rpn_cls=cls, rpn_reg=reg, proposals=props)
        return losses, extras
    return losses


def simple_test(sd, img, rescale=False, scale_factor=(1., 1., 1., 1.), score_thr=0.05, max_per_img=2000):
    """two_stage.py:187-199 -> loft_roi_head.py:196-227 for ONE image (test_cfg of bonai_loft_foa_r50_fpn_basic.py:127-140).
    -> (det_bboxes [n,5], det_labels [n], masks bool [n,H,W], offsets [n,2])."""
    H, W = img.shape[2:]
    feats = extract_feat(sd, img)
    cls, reg = rpn_forward(sd, feats)
    props = rpn_proposals(cls, reg, (H, W, 3))[0]
    rois = R.bbox2roi([props])
    p4 = feats[:4]
    cls_score, bbox_pred = bbox_head(sd, _q(R.roi_extract(p4, rois, 7)))
    scores = torch.softmax(cls_score, dim=1)
    bboxes = R.delta2bbox(rois[:, 1:], bbox_pred, stds=(.1, .1, .2, .2), max_shape=(H, W, 3))
    sf = torch.tensor(scale_factor, dtype=torch.float32)
    if rescale:
        bboxes = bboxes / sf
    det, lab = R.multiclass_nms(bboxes, scores, score_thr, dict(type='soft_nms', iou_threshold=0.5), max_per_img)
    if det.shape[0] == 0:
        return det, lab, torch.zeros(0, H, W, dtype=torch.bool), torch.zeros(0, 2)
    _b = det[:, :4] * sf if rescale else det[:, :4]
    drois = R.bbox2roi([_b])
    mp = mask_head(sd, _q(R.roi_extract(p4, drois, 14))).sigmoid()
    mp = mp[torch.arange(mp.shape[0]), lab][:, None]
    masks = R.paste_masks(mp, _b / sf if rescale else _b, H, W, 0.5)
    op = foa_head(sd, _q(R.roi_extract(p4, drois, 7)))
    offsets = R.delta2offset(_b, R.foa_fuse(op), max_shape=[1024, 1024])
    return det, lab, masks, offsets
