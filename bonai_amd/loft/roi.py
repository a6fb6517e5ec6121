"""RoI side of LOFT on the HIP kernels: SingleRoIExtractor, Shared2FCBBoxHead, FCNMaskHead,
OffsetHead / OffsetHeadExpandFeature (FOA) and LoftRoIHead.

Mirrors (constructor arguments, parameter names, loss keys, output ordering):
  roi_extractors/single_level_roi_extractor.py:9-80     -> one fused multi-level RoIAlign launch
  bbox_heads/convfc_bbox_head.py:176-189, bbox_head.py  -> MFMA GEMMs, fc_cls+fc_reg as one contraction
  mask_heads/fcn_mask_head.py:19-149                    -> tap-conv chain + parity-class deconv
  attribute_heads/offset_head_expand_feature.py:25-461  -> the 4 rotation branches as ONE grouped launch per
        layer; the rotations themselves are written by the RoIAlign kernel (rot90 index permutation)
  attribute_heads/offset_head.py:23-265                 -> plain LOFT head (no FOA)
  loft_roi_head.py:22-227 (+ standard_roi_head.py, test_mixins.py:211-241)
"""
import os

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ..debug import DBG
from .. import nn as F2
from .backbone import ConvW
from .builder import (HEADS, ROI_EXTRACTORS, build_assigner, build_bbox_coder, build_head, build_loss,
                      build_roi_extractor, build_sampler)
from .core import gt_self_inds, pad_gts, pad_rows
from .losses import accuracy


TENSOR_TARGETS = False   # tests: the tensor formulation of the sampled-RoI lists / targets instead of loft_roi_sample_targets
SPECULATIVE_BBOX_ROIALIGN = True   # tests / A-B: the bbox RoIAlign on the worst-case list in front of the count read


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        cfg = dict(roi_layer)
        if cfg.pop('type') != 'RoIAlign' or cfg.get('sampling_ratio', 0) != 0 or cfg.get('pool_mode', 'avg') != 'avg' \
                or not cfg.get('aligned', True):
            raise NotImplementedError('only RoIAlign(sampling_ratio=0, avg, aligned) is built natively')
        self.output_size = int(cfg['output_size'])
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, list(featmap_strides), finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def forward(self, feats, rois, roi_scale_factor=None, n_rot=1):
        if roi_scale_factor is not None:
            raise NotImplementedError('roi_scale_factor')
        return F2.roi_align(list(feats), rois, self.output_size, self.featmap_strides, self.finest_scale, n_rot)


class _FC(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


def _fc_after_flatten(x, fc, relu=True, input_relu=False):
    """``x.flatten(1)`` of an NCHW tensor followed by nn.Linear, on NHWC memory: permute the weight columns
    from (c,y,x) to (y,x,c) order instead of the activations."""
    return F2.linear_after_flatten(x, fc.weight, fc.bias, relu=relu, input_relu=input_relu)


@HEADS.register_module()
class Shared2FCBBoxHead(nn.Module):
    def __init__(self, fc_out_channels=1024, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7,
                 in_channels=256, num_classes=80, bbox_coder=None, reg_class_agnostic=False, reg_decoded_bbox=False,
                 loss_cls=None, loss_bbox=None, **kwargs):
        super().__init__()
        if with_avg_pool or not (with_cls and with_reg) or reg_decoded_bbox:
            raise NotImplementedError('bbox head variant not used by configs/loft_foa')
        self.roi_feat_size, self.in_channels, self.num_classes = roi_feat_size, in_channels, num_classes
        self.reg_class_agnostic = reg_class_agnostic
        self.fc_out_channels = fc_out_channels
        self.bbox_coder = build_bbox_coder(bbox_coder or dict(type='DeltaXYWHBBoxCoder', target_stds=[.1, .1, .2, .2]))
        self.loss_cls = build_loss(loss_cls or dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
        self.loss_bbox = build_loss(loss_bbox or dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
        area = roi_feat_size * roi_feat_size
        # registration order = the reference's (BBoxHead.__init__ creates fc_cls / fc_reg, bbox_head.py:49-56, before
        # ConvFCBBoxHead adds shared_fcs, convfc_bbox_head.py:53-57): model.parameters() order is part of the checkpoint
        # format -- torch.optim.SGD's state_dict indexes parameters by position (tests/test_plugin_cpu.py)
        self.fc_cls = _FC(fc_out_channels, num_classes + 1)
        self.fc_reg = _FC(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)
        self.shared_fcs = nn.ModuleList([_FC(in_channels * area, fc_out_channels), _FC(fc_out_channels, fc_out_channels)])

    def init_weights(self):
        """bbox_head.py:66-73 + convfc_bbox_head.py:126-133."""
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)
        for fc in self.shared_fcs:
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def forward(self, x):
        """x bf16 [N,256,7,7] -> (cls_score fp32 [N,num_classes+1], bbox_pred fp32 [N,4*num_classes])."""
        N = x.shape[0]
        ncls = self.num_classes + 1
        nreg = self.fc_reg.weight.shape[0]
        if N == 0:
            return x.new_zeros(0, ncls, dtype=torch.float32), x.new_zeros(0, nreg, dtype=torch.float32)
        h = _fc_after_flatten(x, self.shared_fcs[0])
        h = F2.linear(h, self.shared_fcs[1].weight, self.shared_fcs[1].bias, relu=True, input_relu=True)
        w = torch.cat([self.fc_cls.weight, self.fc_reg.weight], 0)
        b = torch.cat([self.fc_cls.bias, self.fc_reg.bias], 0)
        o = F2.narrow_head(h.reshape(N, -1, 1, 1).contiguous(memory_format=torch.channels_last), w, b,
                           leaves=[(self.fc_cls.weight, self.fc_cls.bias, 0, ncls),
                                   (self.fc_reg.weight, self.fc_reg.bias, ncls, ncls + nreg)]).reshape(N, -1)
        return o[:, :ncls], o[:, ncls:ncls + nreg]

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights, from_get_targets=False):
        """bbox_head.py:140-185.  from_get_targets: the weights are BBoxHead.get_targets' own (label_weights all one, bbox_weights
        one exactly on the foreground rows, bbox_head.py:84-138) -- the count of valid rows and the foreground mask are then
        known without the nine small launches that recompute them."""
        losses = dict()
        avg = float(max(int(label_weights.shape[0]), 1)) if from_get_targets else (label_weights > 0).sum().float().clamp(min=1.)
        losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg)
        losses['acc'] = accuracy(cls_score, labels, loss_module=self.loss_cls)
        n = bbox_pred.shape[0]
        pred = bbox_pred.view(n, -1, 4)
        if pred.shape[1] == 1:          # one class (BONAI): nothing to select -- no gather / scatter-back launches
            pred = pred[:, 0]
        else:
            idx = labels.clamp(max=pred.shape[1] - 1)
            pred = pred[torch.arange(n, device=pred.device), idx]
        if from_get_targets:
            w = bbox_weights
        else:
            pos = (labels >= 0) & (labels < self.num_classes)
            w = bbox_weights * pos[:, None].float()
        losses['loss_bbox'] = self.loss_bbox(pred, bbox_targets, w, avg_factor=float(max(n, 1)))
        return losses


@HEADS.register_module()
class FCNMaskHead(nn.Module):
    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3, conv_out_channels=256,
                 num_classes=80, class_agnostic=False, upsample_cfg=dict(type='deconv', scale_factor=2), conv_cfg=None,
                 norm_cfg=None, loss_mask=None):
        super().__init__()
        if upsample_cfg.get('type') != 'deconv' or upsample_cfg.get('scale_factor', 2) != 2 or conv_kernel_size != 3 \
                or conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('mask head variant not used by configs/loft_foa')
        self.num_convs, self.num_classes, self.class_agnostic = num_convs, num_classes, class_agnostic
        self.loss_mask = build_loss(loss_mask or dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))

        class _CM(nn.Module):
            def __init__(self, cin, cout):
                super().__init__()
                self.conv = ConvW(cin, cout, 3, bias=True)
        self.convs = nn.ModuleList([_CM(in_channels if i == 0 else conv_out_channels, conv_out_channels)
                                    for i in range(num_convs)])
        self.upsample = nn.Module()
        self.upsample.weight = nn.Parameter(torch.empty(conv_out_channels, conv_out_channels, 2, 2))
        self.upsample.bias = nn.Parameter(torch.zeros(conv_out_channels))
        self.conv_logits = ConvW(conv_out_channels, 1 if class_agnostic else num_classes, 1, bias=True)

    def init_weights(self):
        """ConvModule default init for convs (kaiming, relu); fcn_mask_head.py:107-116 for the rest."""
        for m in self.convs:
            nn.init.kaiming_normal_(m.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.conv.bias, 0)
        for m in (self.upsample, self.conv_logits):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        """x bf16 [N,256,14,14] -> mask logits fp32 [N,num_classes,28,28]."""
        nout = self.conv_logits.weight.shape[0]
        if x.shape[0] == 0:
            return x.new_zeros(0, nout, 2 * x.shape[2], 2 * x.shape[3], dtype=torch.float32)
        for i, m in enumerate(self.convs):
            x = F2.conv2d(x, m.conv.weight, m.conv.bias, pad=1, relu=True, input_relu=i > 0)
        # the logits ride in the deconvolution's epilogue (fcn_mask_head.py:121-126: upsample -> relu -> conv_logits): the
        # [N,256,28,28] map is written once and not read back by a launch of its own
        wl = self.conv_logits.weight.view(nout, -1)
        pre = F2.narrow_head_prepack(wl, self.conv_logits.bias, x.dtype) if x.dtype == K.L.act16() else None
        x, o_pre = F2.deconv2x2_relu(x, self.upsample.weight, self.upsample.bias, input_relu=len(self.convs) > 0, head=pre) \
            if pre is not None else (F2.deconv2x2_relu(x, self.upsample.weight, self.upsample.bias, input_relu=len(self.convs) > 0), None)
        o = F2.narrow_head(x, wl, self.conv_logits.bias, input_relu=True, prepacked=pre, precomputed=o_pre,
                           leaves=[(self.conv_logits.weight, self.conv_logits.bias, 0, nout)])
        return o[:, :nout]

    def loss(self, mask_pred, mask_targets, labels):
        if mask_pred.size(0) == 0:
            return dict(loss_mask=mask_pred.sum() * 0)
        if self.class_agnostic:
            labels = torch.zeros_like(labels)
        return dict(loss_mask=self.loss_mask(mask_pred, mask_targets, labels))


class _OffsetBase(nn.Module):
    def _common(self, roi_feat_size, in_channels, conv_out_channels, fc_out_channels, num_fcs, reg_num, offset_coder,
                loss_offset, offset_coordinate, reg_decoded_offset):
        if reg_num not in (2, 3) or offset_coordinate not in ('rectangle', 'polar') or num_fcs < 1:
            raise NotImplementedError('offset head variant not used by configs/loft_foa')
        if getattr(self, 'expand_feature_num', 1) != 1 and (reg_num != 2 or offset_coordinate != 'rectangle' or reg_decoded_offset):
            raise NotImplementedError('FOA is built for reg_num=2 rectangular encoded offsets (configs/loft_foa)')
        self.offset_coordinate, self.reg_decoded_offset = offset_coordinate, reg_decoded_offset
        self.roi_feat_size, self.in_channels, self.conv_out_channels = roi_feat_size, in_channels, conv_out_channels
        self.fc_out_channels, self.reg_num = fc_out_channels, reg_num
        self.offset_coder = build_bbox_coder(offset_coder)
        self.loss_offset = build_loss(loss_offset)
        area = roi_feat_size * roi_feat_size

        def stack():
            return nn.ModuleList([_FC(conv_out_channels * area if i == 0 else fc_out_channels, fc_out_channels)
                                  for i in range(num_fcs)])
        if getattr(self, 'share_expand_fc', True):
            self.fcs = stack()
            self.fc_offset = _FC(fc_out_channels, reg_num)
        else:
            # one FC stack and one regressor PER rotation branch (offset_head_expand_feature.py:82-95; the class default)
            self.expand_fcs = nn.ModuleList([stack() for _ in range(self.expand_feature_num)])
            self.expand_fc_offsets = nn.ModuleList([_FC(fc_out_channels, reg_num) for _ in range(self.expand_feature_num)])

    def _init_fcs(self):
        shared = getattr(self, 'share_expand_fc', True)
        for fc in (self.fcs if shared else [fc for fcs in self.expand_fcs for fc in fcs]):
            nn.init.kaiming_uniform_(fc.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
            nn.init.constant_(fc.bias, 0)
        for fo in ([self.fc_offset] if shared else self.expand_fc_offsets):
            nn.init.normal_(fo.weight, 0, 0.01)
            nn.init.constant_(fo.bias, 0)

    def _fc_tail(self, x, fcs=None, fc_offset=None):
        fcs = self.fcs if fcs is None else fcs
        fc_offset = self.fc_offset if fc_offset is None else fc_offset
        N = x.shape[0]
        h = _fc_after_flatten(x, fcs[0], input_relu=self.num_convs > 0)   # x: output of the conv + ReLU chain
        for fc in list(fcs)[1:]:
            h = F2.linear(h, fc.weight, fc.bias, relu=True, input_relu=True)
        o = F2.narrow_head(h.reshape(N, -1, 1, 1).contiguous(memory_format=torch.channels_last), fc_offset.weight,
                           fc_offset.bias)
        return o.reshape(N, -1)[:, :self.reg_num]

    def loss(self, offset_pred, offset_targets):
        if offset_pred.size(0) == 0:
            return dict(loss_offset=offset_pred.sum() * 0)
        return dict(loss_offset=self.loss_offset(offset_pred, offset_targets))


class _BranchSlabs(torch.autograd.Function):
    """[E N,C,H,W] branch-major NHWC batch -> its E [N,C,H,W] slabs (views); the backward concatenates the slab
    gradients back into ONE NHWC tensor (autograd's own slice backward would build E full-size NCHW-strided zero tensors)."""

    @staticmethod
    def forward(ctx, x4, E=4):
        n = x4.shape[0] // E
        return tuple(x4[k * n:(k + 1) * n] for k in range(E))

    @staticmethod
    def backward(ctx, *gs):
        return torch.cat(gs, 0).contiguous(memory_format=torch.channels_last), None


class _PickSlabs(torch.autograd.Function):
    """[4N,C,H,W] (the four rot90 copies the RoIAlign kernel writes) -> the slabs `idx` of it, branch-major [len(idx) N,C,H,W]:
    a FOA head with fewer than four rotation branches (offset_head_expand_feature.py:371-385)."""

    @staticmethod
    def forward(ctx, x4, idx):
        n = x4.shape[0] // 4
        ctx.idx, ctx.shape = idx, tuple(x4.shape)
        return torch.cat([x4[k * n:(k + 1) * n] for k in idx], 0).contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        n = ctx.shape[0] // 4
        g4 = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
        for j, k in enumerate(ctx.idx):
            g4[k * n:(k + 1) * n] = g[j * n:(j + 1) * n]
        return g4, None


# rotation sets of the reference's offset_fusion (offset_head_expand_feature.py:371-397); the four-branch set is configs/loft_foa's
FOA_ROTATION_SETS = ([0, 90, 180, 270], [0, 180], [0, 90], [0, 90, 180])


@HEADS.register_module()
class OffsetHeadExpandFeature(_OffsetBase):
    """FOA: rotated copies of the RoI feature (4 in configs/loft_foa; 2 or 3 in the reference's other rotation sets), per-branch
    num_convs x (conv3x3+ReLU), then the FC stack + regressor -- shared by the branches (``share_expand_fc=True``,
    configs/loft_foa) or one per branch (the reference class's default)."""

    def __init__(self, roi_feat_size=7, in_channels=256, num_convs=4, num_fcs=2, reg_num=2, conv_out_channels=256,
                 fc_out_channels=1024, expand_feature_num=4, share_expand_fc=False, rotations=[0, 90, 180, 270],
                 offset_coordinate='rectangle', offset_coder=dict(type='DeltaXYOffsetCoder', target_means=[0.0, 0.0],
                                                                    target_stds=[0.5, 0.5]),
                 reg_decoded_offset=False, conv_cfg=None, norm_cfg=None, loss_offset=dict(type='MSELoss', loss_weight=1.0)):
        super().__init__()
        if list(rotations) not in FOA_ROTATION_SETS or expand_feature_num != len(rotations) or in_channels != conv_out_channels:
            raise NotImplementedError('FOA takes the rotation sets the reference\'s offset_fusion names: '
                                      '[0,90,180,270], [0,180], [0,90], [0,90,180] (offset_head_expand_feature.py:371-397)')
        self.expand_feature_num, self.rotations, self.share_expand_fc = expand_feature_num, list(rotations), bool(share_expand_fc)
        self.rot_idx = tuple(r // 90 for r in self.rotations)           # slab of the RoIAlign kernel's four rot90 copies per branch
        self.num_convs = num_convs
        if tuple(float(v) for v in offset_coder.get('target_means', (0., 0.))) != (0., 0.):
            raise NotImplementedError('FOA target / fusion kernels take zero offset means (configs/loft_foa)')
        # (convs before fcs / fc_offset: the reference's registration order, offset_head_expand_feature.py:62-107)
        self.expand_convs = nn.ModuleList([nn.ModuleList([ConvW(conv_out_channels, conv_out_channels, 3, bias=True)
                                                          for _ in range(num_convs)]) for _ in range(expand_feature_num)])
        self._common(roi_feat_size, in_channels, conv_out_channels, fc_out_channels, num_fcs, reg_num, offset_coder,
                     loss_offset, offset_coordinate, reg_decoded_offset)

    def init_weights(self):
        """offset_head_expand_feature.py:109-132."""
        for convs in self.expand_convs:
            for c in convs:
                nn.init.kaiming_normal_(c.weight, a=0, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(c.bias, 0)
        self._init_fcs()

    def forward_rotated(self, x4):
        """x4 bf16 [4N,256,7,7], branch-major (the RoIAlign kernel already wrote the 4 rotations)
        -> fp32 [E N,2] branch-major (= torch.cat(offsets, 0) of offset_head_expand_feature.py:160)."""
        E = self.expand_feature_num
        if x4.shape[0] == 0:
            return x4.new_zeros(0, 2 * E, dtype=torch.float32)  # appendix A.1 quirk
        if E != 4:
            x4 = _PickSlabs.apply(x4, self.rot_idx)
        for i in range(self.num_convs):
            x4 = F2.conv2d(x4, [self.expand_convs[k][i].weight for k in range(E)],
                           [self.expand_convs[k][i].bias for k in range(E)], pad=1, relu=True, groups=E, input_relu=i > 0)
        if self.share_expand_fc:
            return self._fc_tail(x4)                   # one [E N, .] GEMM per shared layer
        # share_expand_fc=False (offset_head_expand_feature.py:147-152): branch k's rows through branch k's own FCs; the rows of
        # one branch are a contiguous slab of the branch-major batch
        return torch.cat([self._fc_tail(xk, self.expand_fcs[k], self.expand_fc_offsets[k])
                          for k, xk in enumerate(_BranchSlabs.apply(x4, E))], 0)

    def forward(self, x):
        """Reference signature: un-rotated RoI features [N,256,7,7] in, [E N,2] out."""
        x4 = torch.cat([torch.rot90(x, k, (2, 3)) for k in range(4)], 0).contiguous(memory_format=torch.channels_last)
        return self.forward_rotated(x4)

    def get_targets(self, pos_bboxes, pos_gt_offsets):
        t4 = K.foa_targets(pos_bboxes, pos_gt_offsets, self.offset_coder.stds)
        if self.expand_feature_num == 4:
            return t4
        n = t4.shape[0] // 4
        return torch.cat([t4[k * n:(k + 1) * n] for k in self.rot_idx], 0)

    def _as_four(self, offset_pred):
        """[E N,2] -> the [4N,2] the four-branch fusion kernel takes, such that its 'max' is the E-branch one: a missing rotation
        slot repeats the main branch (with the columns swapped where the kernel swaps them back), which a maximum ignores."""
        E = self.expand_feature_num
        if E == 4:
            return offset_pred
        n = offset_pred.shape[0] // E
        b = offset_pred.split(n, 0)
        slots = []
        for k in range(4):
            if k in self.rot_idx:
                slots.append(b[self.rot_idx.index(k)])
            else:
                slots.append(b[0][:, [1, 0]] if k % 2 else b[0])
        return torch.cat(slots, 0).contiguous()

    def offset_fusion(self, offset_pred, model='max'):
        """offset_head_expand_feature.py:346-413 on the device: [E N,2] -> fused [N,2] encoded offsets.  'max' = what get_offsets
        uses; 'mean' = the SUM of the branches' magnitudes (the reference divides by 1).  Sign of the main branch, zero -> -1."""
        E = self.expand_feature_num
        n = offset_pred.shape[0] // E
        b = offset_pred.float().split(n, 0)
        cur = [b[i][:, [1, 0]] if self.rotations[i] in (90, 270) else b[i] for i in range(E)]
        mags = torch.stack([c.abs() for c in cur], 0)
        if model == 'max':
            val = mags.max(0)[0]
        elif model == 'mean':
            val = mags.sum(0)
        else:
            raise NotImplementedError(model)
        return val * torch.where(b[0] > 0, torch.ones_like(b[0]), -torch.ones_like(b[0]))

    def get_offsets(self, offset_pred, det_bboxes, scale_factor=None, rescale=False, img_shape=(1024, 1024)):
        return K.foa_fuse_decode(self._as_four(offset_pred), det_bboxes, self.offset_coder.stds,
                                 img_shape).cpu().numpy().astype(np.float32)


@HEADS.register_module()
class OffsetHead(_OffsetBase):
    """Basic LOFT offset head without FOA (attribute_heads/offset_head.py:23-265)."""

    def __init__(self, roi_feat_size=7, in_channels=256, num_convs=4, num_fcs=2, reg_num=2, conv_out_channels=256,
                 fc_out_channels=1024, offset_coordinate='rectangle', offset_coder=dict(
                     type='DeltaXYOffsetCoder', target_means=[0.0, 0.0], target_stds=[0.5, 0.5]),
                 reg_decoded_offset=False, conv_cfg=None, norm_cfg=None, loss_offset=dict(type='MSELoss', loss_weight=1.0)):
        super().__init__()
        self.num_convs = num_convs
        self.convs = nn.ModuleList([ConvW(in_channels if i == 0 else conv_out_channels, conv_out_channels, 3, bias=True)
                                    for i in range(num_convs)])
        self._common(roi_feat_size, in_channels, conv_out_channels, fc_out_channels, num_fcs, reg_num, offset_coder,
                     loss_offset, offset_coordinate, reg_decoded_offset)

    def init_weights(self):
        for c in self.convs:
            nn.init.kaiming_normal_(c.weight, a=0, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(c.bias, 0)
        self._init_fcs()

    def forward(self, x):
        """bf16 NHWC RoI features [N,C,7,7] -> fp32 [N,reg_num] (offset_head.py:90-106; empty input -> (0, 2))."""
        if x.shape[0] == 0:
            return x.new_zeros(0, 2, dtype=torch.float32)
        for i, c in enumerate(self.convs):
            x = F2.conv2d(x, c.weight, c.bias, pad=1, relu=True, input_relu=i > 0)
        return self._fc_tail(x)

    def get_targets(self, pos_bboxes, pos_gt_offsets):
        """offset_head.py:118-188 on the concatenated positives: [n,4] boxes + their assigned gt offsets [n,2] (what
        `_offset_target_single`'s per-RoI python loop gathers) -> [n,reg_num]; one launch."""
        if self.reg_decoded_offset:
            t = pos_gt_offsets.float()
            return t if self.reg_num == 2 else torch.stack([t[:, 0], torch.cos(t[:, 1]), torch.sin(t[:, 1])], -1)
        return K.offset_targets(pos_bboxes, pos_gt_offsets, self.offset_coder.means, self.offset_coder.stds, self.reg_num)

    def get_offsets(self, offset_pred, det_bboxes, scale_factor=None, rescale=False, img_shape=(1024, 1024)):
        """offset_head.py:190-243 -> np.float32 [n,2] (pixels of the network input; scale_factor is ignored there too)."""
        o = K.offset_decode(offset_pred, det_bboxes, self.offset_coder.means, self.offset_coder.stds, img_shape,
                            polar=self.offset_coordinate == 'polar')
        return o.cpu().numpy().astype(np.float32)


def _masks_to_device(gt_masks, device):
    """list of BitmapMasks-like (``.masks`` ndarray [K,H,W]) / ndarrays / uint8 tensors -> (list of uint8 device tensors
    [K_b,H,W], instance offsets); the mask-target kernel addresses the instances in place (no concatenation)."""
    ts = []
    for m in gt_masks:
        a = getattr(m, 'masks', m)
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        ts.append(t.to(device=device, dtype=torch.uint8))
    offs = [0]
    for t in ts:
        offs.append(offs[-1] + int(t.shape[0]))
    return ts, offs


@HEADS.register_module()
class LoftRoIHead(nn.Module):
    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 offset_roi_extractor=None, offset_head=None, shared_head=None, train_cfg=None, test_cfg=None):
        super().__init__()
        assert offset_head is not None and bbox_head is not None
        if shared_head is not None:
            raise NotImplementedError('shared_head')
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_head(bbox_head)
        self.with_mask = mask_head is not None
        if self.with_mask:
            self.share_roi_extractor = mask_roi_extractor is None
            self.mask_roi_extractor = self.bbox_roi_extractor if mask_roi_extractor is None else build_roi_extractor(
                mask_roi_extractor)
            self.mask_head = build_head(mask_head)
        self.offset_roi_extractor = build_roi_extractor(offset_roi_extractor)
        self.offset_head = build_head(offset_head)
        self.with_vis_feat = False
        if train_cfg is not None:
            self.bbox_assigner = build_assigner(train_cfg.assigner)
            self.bbox_sampler = build_sampler(train_cfg.sampler)
        self.last_stats = {}

    with_bbox = True
    with_offset = True

    def init_weights(self, pretrained=None):
        self.bbox_head.init_weights()
        if self.with_mask:
            self.mask_head.init_weights()
        self.offset_head.init_weights()

    # ---------------------------------------------------------------- training
    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                      gt_offsets=None):
        """proposal_list: (proposals [B,P,5], counts [B]) from RPNHead, or the reference's list of (n_i,5) tensors."""
        dev = x[0].device
        B = len(img_metas)
        if isinstance(proposal_list, (list,)):
            P = max(1, max(int(p.shape[0]) for p in proposal_list))
            props = torch.zeros(B, P, 5, device=dev)
            for i, p in enumerate(proposal_list):
                props[i, :p.shape[0], :p.shape[1]] = p
            nprop = K.h2d([int(p.shape[0]) for p in proposal_list], torch.int64, dev)
        else:
            props, nprop = proposal_list
        with torch.no_grad():
            gts, ngt = pad_gts(gt_bboxes, dev)
            Kmax = gts.shape[1]
            gi_p, _ = self.bbox_assigner.assign_batched(props[..., :4].contiguous(), nprop.int(), gts, ngt)
            if self.bbox_sampler.add_gt_as_proposals:
                gi_g = gt_self_inds(gt_bboxes, dev)
                gt_inds = torch.cat([gi_g, gi_p], 1)
                cand = torch.cat([gts, props[..., :4]], 1)
            else:
                gt_inds, cand = gi_p, props[..., :4]
            smp = self.bbox_sampler.sample_batched(gt_inds)
            pidx, pval, nidx, nval = smp['pos_idx'], smp['pos_valid'], smp['neg_idx'], smp['neg_valid']
            lab_pad = pad_rows(gt_labels, dev, torch.long, Kmax)
        fused_targets = dev.type == 'cuda' and not TENSOR_TARGETS
        spec_feats = None
        if fused_targets:
            # per image [pos..., neg...] (SamplingResult.bboxes, sampling_result.py:50-53), labels and bbox targets: one launch
            with torch.no_grad():
                pending = K.roi_sample_targets_begin(cand, gt_inds, gts, lab_pad, pidx, pval, nidx, nval,
                                                     self.bbox_head.num_classes, self.bbox_head.bbox_coder.means,
                                                     self.bbox_head.bbox_coder.stds, num_expected=self.bbox_sampler.num)
            # The bbox extractor runs on the worst-case RoI list BEFORE the host learns the counts: the GPU has ~0.3 ms of work
            # while the host waits for them and then enqueues the branches (the read used to leave the device idle).  Every
            # sampler slot is filled in all but degenerate batches; otherwise the result is dropped and recomputed below.
            if SPECULATIVE_BBOX_ROIALIGN and pending.rois_max.shape[0] > 0:
                xb_ = x.branches[1] if isinstance(x, F2.FeatFork) else x
                spec_feats = self.bbox_roi_extractor(xb_[:self.bbox_roi_extractor.num_inputs], pending.rois_max)
            tg = pending.finish()
            rois, labels, label_weights = tg['rois'], tg['labels'], tg['label_weights']
            bbox_targets, bbox_weights = tg['bbox_targets'], tg['bbox_weights']
            pos_rois, pos_b, pos_gt_i, pos_sel = tg['pos_rois'], tg['pos_b'], tg['pos_gt_i'], tg['pos_sel']
            M = rois.shape[0]
        with torch.no_grad():
          if not fused_targets:
              # per image [pos..., neg...] (SamplingResult.bboxes, sampling_result.py:50-53); two host syncs (nonzero)
              idx = torch.cat([pidx, nidx], 1)
              val = torch.cat([pval, nval], 1)
              is_pos = torch.cat([pval, torch.zeros_like(nval)], 1)
              bidx = torch.arange(B, device=dev)[:, None].expand_as(idx)
              sel = val.reshape(-1).nonzero(as_tuple=False).flatten()
              b_s, i_s, pos_s = bidx.reshape(-1)[sel], idx.reshape(-1)[sel], is_pos.reshape(-1)[sel]
              boxes_s = cand[b_s, i_s]
              rois = torch.cat([b_s[:, None].float(), boxes_s], 1)
              assigned = (gt_inds[b_s, i_s] - 1).clamp(min=0)
              pos_sel = pos_s.nonzero(as_tuple=False).flatten()
              pos_rois, pos_b, pos_gt_i = rois[pos_sel], b_s[pos_sel], assigned[pos_sel]
              pos_gt_boxes = gts[pos_b, pos_gt_i]
              M = rois.shape[0]
              labels = torch.full((M,), self.bbox_head.num_classes, dtype=torch.long, device=dev)
              labels[pos_sel] = lab_pad[pos_b, pos_gt_i]
              bbox_targets = torch.zeros(M, 4, device=dev)
              bbox_weights = torch.zeros(M, 4, device=dev)
              if pos_sel.numel():
                  bbox_targets[pos_sel] = self.bbox_head.bbox_coder.encode(pos_rois[:, 1:].contiguous(), pos_gt_boxes)
                  bbox_weights[pos_sel] = 1.0
              label_weights = torch.ones(M, device=dev)
        self.last_stats = dict(num_rois=int(M), num_pos=int(pos_sel.numel()))

        losses = dict()
        xb = xm = xo = x
        if isinstance(x, F2.FeatFork):       # own aliases per extractor: their backward kernels share one gradient map per level
            xb, xm, xo = x.branches[1], x.branches[2], x.branches[3]
        feats = xb[:self.bbox_roi_extractor.num_inputs]
        if spec_feats is not None and spec_feats.shape[0] == rois.shape[0]:
            bbox_feats = spec_feats
        else:
            if spec_feats is not None:           # under-filled sampler: the speculative node is dropped, never differentiated
                F2.roi_align_discard(spec_feats)
                spec_feats = None
            bbox_feats = self.bbox_roi_extractor(feats, rois)

        def bbox_branch():
            cls_score, bbox_pred = self.bbox_head(bbox_feats)
            return self.bbox_head.loss(cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
                                       from_get_targets=fused_targets)
        bbox_on_side = self.with_mask and not DBG.no_bbox_side_stream   # the bbox head's 512-workgroup GEMMs ride along
        if not bbox_on_side:
            losses.update(bbox_branch())

        side = None
        if self.with_mask:
            # The mask branch (4 convs + deconv + logits at 14x14 / 28x28) and the FOA branch (40 convs at 7x7) are independent
            # chains of launches that each fill 2.6 rounds of the 256 CUs: on two HIP streams their tails fill each other's
            # idle CUs, forward and (autograd replays each node on its forward stream) backward.  RoIAlign stays on the main
            # stream -- its backward accumulates into the shared per-level gradient maps.
            mask_losses = dict()
            if dev.type == 'cuda' and torch.is_grad_enabled() and K.PROFILE is None and not DBG.no_side_stream:
                if getattr(self, '_side_stream', None) is None:
                    self._side_stream = torch.cuda.Stream()
                side = self._side_stream
                if bbox_on_side:
                    # the bbox head forks right behind ITS RoIAlign: its two FC GEMMs and loss launches (~0.3 ms) run beside the
                    # mask extractor's RoIAlign (0.28 ms of dependent loads that leave the matrix pipes idle) instead of after it
                    side.wait_stream(torch.cuda.current_stream())
                    bbox_feats.record_stream(side)
                    with torch.cuda.stream(side):
                        mask_losses.update(bbox_branch())
            mask_feats = self.mask_roi_extractor(xm[:self.mask_roi_extractor.num_inputs], pos_rois)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                mask_feats.record_stream(side)

            def mask_branch():
                mask_pred = self.mask_head(mask_feats)
                with torch.no_grad():
                    masks, moffs = _masks_to_device(gt_masks, dev)
                    H, W = masks[0].shape[1], masks[0].shape[2]
                    pb = pos_rois[:, 1:].clone()
                    pb[:, 0::2].clamp_(0, W)      # (strided views: list indices would cost two H2D copies and six launches)
                    pb[:, 1::2].clamp_(0, H)
                    gidx = pos_gt_i + K.h2d(moffs[:-1], torch.int64, dev)[pos_b]
                    mask_targets = K.mask_target(masks, pb, gidx, int(self.train_cfg.mask_size))
                return self.mask_head.loss(mask_pred, mask_targets, labels[pos_sel])
            if side is not None:
                with torch.cuda.stream(side):
                    mask_losses.update(mask_branch())
            else:
                if bbox_on_side:
                    losses.update(bbox_branch())
                losses.update(mask_branch())

        with torch.no_grad():
            off_pad = pad_rows(gt_offsets, dev, torch.float32, Kmax)
            pos_gt_off = off_pad[pos_b, pos_gt_i]
        offset_pred = self._offset_forward(xo, pos_rois)
        with torch.no_grad():
            offset_targets = self.offset_head.get_targets(pos_rois[:, 1:].contiguous(), pos_gt_off)
        if offset_pred.shape[0] == 0:
            losses.update(loss_offset=offset_pred.sum() * 0)
        else:
            losses.update(self.offset_head.loss(offset_pred, offset_targets))
        if side is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            for v in mask_losses.values():
                v.record_stream(main)
            losses.update(mask_losses)
        return losses

    # ---------------------------------------------------------------- inference
    def multiclass_nms(self, multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
        """mmdet/core/post_processing/bbox_nms.py:5-69 on the device (soft-NMS included)."""
        num_classes = multi_scores.size(1) - 1
        if multi_bboxes.shape[1] > 4:
            bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
        else:
            bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
        scores = multi_scores[:, :-1]
        valid = scores > score_thr
        bboxes = bboxes[valid]
        scores = scores[valid]
        labels = valid.nonzero(as_tuple=False)[:, 1]
        if bboxes.numel() == 0:
            return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
        dets, keep = K.batched_nms(bboxes.contiguous(), scores.contiguous(), labels, nms_cfg)
        if max_num > 0:
            dets, keep = dets[:max_num], keep[:max_num]
        return dets, labels[keep]

    @torch.no_grad()
    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        """loft_roi_head.py:196-227 (+ test_mixins.py:53-72,152-177,213-241): one image ->
        (bbox_results, segm_results, offset_results) with the reference's types."""
        dev = x[0].device
        if isinstance(proposal_list, (list,)):
            props = proposal_list[0]
        else:
            p, cnt = proposal_list
            props = p[0, :int(cnt[0])]
        meta = img_metas[0]
        img_shape, scale_factor, ori_shape = meta['img_shape'], meta['scale_factor'], meta['ori_shape']
        rois = torch.cat([props.new_zeros(props.shape[0], 1), props[:, :4]], 1)
        cfg = self.test_cfg
        cls_score, bbox_pred = self.bbox_head(self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois))
        scores = torch.softmax(cls_score, dim=1)
        ncls = self.bbox_head.num_classes
        bp = bbox_pred.reshape(-1, 4)
        rr = rois[:, 1:].repeat_interleave(bbox_pred.shape[1] // 4, dim=0)
        bboxes = self.bbox_head.bbox_coder.decode(rr, bp, max_shape=img_shape).view(rois.shape[0], -1)
        sf = torch.as_tensor(np.asarray(scale_factor, dtype=np.float32), device=dev)
        if rescale:
            bboxes = (bboxes.view(bboxes.size(0), -1, 4) / sf).view(bboxes.size(0), -1)
        det_bboxes, det_labels = self.multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)
        db = det_bboxes.cpu().numpy()
        dl = det_labels.cpu().numpy()
        bbox_results = [db[dl == i, :] for i in range(ncls)] if db.shape[0] else \
            [np.zeros((0, 5), dtype=np.float32) for _ in range(ncls)]
        if det_bboxes.shape[0] == 0:
            segm = [[] for _ in range(ncls)] if self.with_mask else None
            return bbox_results, segm, [[] for _ in range(2)]
        _bboxes = det_bboxes[:, :4] * sf if rescale else det_bboxes[:, :4]
        det_rois = torch.cat([_bboxes.new_zeros(_bboxes.shape[0], 1), _bboxes], 1).contiguous()
        segm_results = None
        if self.with_mask:
            mask_pred = self.mask_head(self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], det_rois))
            sel = mask_pred[torch.arange(mask_pred.shape[0], device=dev), 0 if self.mask_head.class_agnostic else det_labels]
            if rescale:
                img_h, img_w = ori_shape[:2]
                pb = _bboxes / sf
            else:
                img_h = int(np.round(ori_shape[0] * float(np.asarray(scale_factor).reshape(-1)[1 if np.size(scale_factor) > 1 else 0])))
                img_w = int(np.round(ori_shape[1] * float(np.asarray(scale_factor).reshape(-1)[0])))
                pb = _bboxes
            pasted = K.mask_paste(sel, pb.contiguous(), img_h, img_w, cfg.mask_thr_binary)
            segm_results = [[] for _ in range(ncls)]
            if cfg.get('rle_masks', False):
                # what apis/test.py:59-67 (encode_mask_results) produces from the bool arrays, without moving them to the host:
                # run boundaries are extracted on the device (bonai_amd.rle)
                from ..rle import rle_encode_masks
                for i, r in enumerate(rle_encode_masks(pasted)):
                    segm_results[int(dl[i])].append(r)
            else:
                im = pasted.bool().cpu().numpy()
                for i in range(im.shape[0]):
                    segm_results[int(dl[i])].append(im[i])
        offset_pred = self._offset_forward(x, det_rois)
        offset_results = self.offset_head.get_offsets(offset_pred, _bboxes.contiguous(), scale_factor, rescale)
        if cfg.get('keep_device_masks', False):
            # tools/test.py --eval: the pasted roof bitmaps stay on the device for bonai_amd.evaluation (footprints by
            # kernels.mask_translate, IoU pairing) -- next to, not instead of, the result tuple
            self.last_device_masks = pasted if self.with_mask else None
            self.last_dets, self.last_det_labels = db, dl              # detection order = the order of the bitmaps and offsets
        return bbox_results, segm_results, offset_results

    def forward_dummy(self, x, proposals):
        """standard_roi_head.py:54-68: (cls_score, bbox_pred, mask_pred of the first 100 RoIs)."""
        rois = torch.cat([proposals.new_zeros(proposals.shape[0], 1), proposals[:, :4]], 1).contiguous()
        outs = tuple(self.bbox_head(self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)))
        if self.with_mask:
            outs = outs + (self.mask_head(self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], rois[:100].contiguous())),)
        return outs

    def _offset_forward(self, x, rois):
        feats = x[:self.offset_roi_extractor.num_inputs]
        if isinstance(self.offset_head, OffsetHeadExpandFeature):
            return self.offset_head.forward_rotated(self.offset_roi_extractor(feats, rois, n_rot=4))
        return self.offset_head(self.offset_roi_extractor(feats, rois))
