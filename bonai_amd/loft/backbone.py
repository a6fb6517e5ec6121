"""ResNet-50 backbone + FPN neck on the MFMA tap-conv kernels.

Mirrors the constructor surface and parameter names of the reference's
``ResNet`` (mmdet/models/backbones/resnet.py:303-649; Bottleneck :95-300; ResLayer
models/utils/res_layer.py:5-102) and ``FPN`` (mmdet/models/necks/fpn.py:64-216) so
``configs/_base_/models/bonai_loft_foa_r50_fpn_basic.py:5-18`` builds unchanged and
torchvision / reference checkpoints load by key.  Differences in *how*: NHWC bf16 activations,
frozen-statistics BatchNorm folded into the conv weights each step (tools/fuse_conv_bn.py:10-23
formula, kept differentiable w.r.t. gamma/beta), BN shift + residual add + ReLU applied in the
conv epilogue, and the stem as one fused kernel.
"""
import os

import torch
from torch import nn

from .. import kernels as K
from ..debug import DBG
from .. import nn as F2
from .builder import BACKBONES, NECKS


class FrozenStatBN(nn.Module):
    """BatchNorm2d in ``norm_eval`` mode: running statistics fixed, affine (gamma, beta) trainable."""

    def __init__(self, num_features, eps=1e-5, requires_grad=True):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features), requires_grad=requires_grad)
        self.bias = nn.Parameter(torch.zeros(num_features), requires_grad=requires_grad)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def fold(self):
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias - self.running_mean * scale


class ConvW(nn.Module):
    """Holds ``weight`` (and optional ``bias``) under the reference's names; no compute of its own."""

    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


class ModulatedDeformConvPack(nn.Module):
    """Parameter holder with the names of mmcv's ModulatedDeformConv2dPack [mmcv==1.0.5]: ``weight`` [Cout,Cin,k,k],
    optional ``bias``, ``conv_offset.{weight,bias}`` [3*DG*k*k, Cin, k, k].  Compute: bonai_amd.nn.modulated_deform_conv2d
    (conv_offset -> loft_mdcn_sample -> 1x1 MFMA contraction).  Built where the reference builds conv type 'DCNv2'
    (resnet.py:171-194; fpn.py:116-132 through ConvModule's conv_cfg)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, deform_groups=1, bias=True):
        super().__init__()
        self.k, self.stride, self.padding, self.deform_groups = k, stride, padding, deform_groups
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.conv_offset = ConvW(cin, deform_groups * 3 * k * k, k, bias=True)
        self.init_weights()

    def init_weights(self):
        """mmcv ModulatedDeformConv2d.init_weights: U(-1/sqrt(fan_in), +), zero bias; conv_offset zero (resnet.py:608-612)."""
        stdv = 1.0 / (self.weight.shape[1] * self.k * self.k) ** 0.5
        nn.init.uniform_(self.weight, -stdv, stdv)
        if self.bias is not None:
            nn.init.zeros_(self.bias)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x, bn=None, relu=False):
        return F2.modulated_deform_conv2d(x, self.weight, self.bias, self.conv_offset.weight, self.conv_offset.bias,
                                          stride=self.stride, pad=self.padding, deform_groups=self.deform_groups, bn=bn,
                                          relu=relu)


def _dcn_groups(dcn):
    if dcn.get('type') != 'DCNv2':
        raise NotImplementedError(f"conv type {dcn.get('type')!r}: only DCNv2 (modulated) has a native kernel")
    return dcn.get('deform_groups', dcn.get('deformable_groups', 1))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False, bn_grad=True, dcn=None):
        super().__init__()
        self.conv1 = ConvW(inplanes, planes, 1)
        self.bn1 = FrozenStatBN(planes, requires_grad=bn_grad)
        if dcn is not None and not (dcn.get('fallback_on_stride', False) and stride > 1):   # resnet.py:171-194
            self.conv2 = ModulatedDeformConvPack(planes, planes, 3, stride=stride, padding=1,
                                                 deform_groups=_dcn_groups(dcn), bias=False)
        else:
            self.conv2 = ConvW(planes, planes, 3)
        self.bn2 = FrozenStatBN(planes, requires_grad=bn_grad)
        self.conv3 = ConvW(planes, planes * 4, 1)
        self.bn3 = FrozenStatBN(planes * 4, requires_grad=bn_grad)
        self.stride = stride                     # style='pytorch': the stride sits on the 3x3 (resnet.py:151-153)
        if downsample:
            self.downsample = nn.Sequential()
            self.downsample.add_module('0', ConvW(inplanes, planes * 4, 1))
            self.downsample.add_module('1', FrozenStatBN(planes * 4, requires_grad=bn_grad))
        else:
            self.downsample = None

    def one_node(self):
        """This block runs as ONE autograd node (nn.res_block)?  (DBG.no_block_fusion: A/B switch -- one node per conv instead)"""
        return not DBG.no_block_fusion and not isinstance(self.conv2, ModulatedDeformConvPack)

    def forward(self, x, pair=None):
        """pair: the hand-over dict of the stage's consecutive blocks (nn.res_block: pair fusion of this block's conv3 with the next
        block's conv1, loft_bneck_pair_bf16); None = every conv a launch of its own."""
        if self.one_node():
            main = [(self.conv1.weight, self.bn1, 1, 1, 0, None), (self.conv2.weight, self.bn2, 3, self.stride, 1, None),
                    (self.conv3.weight, self.bn3, 1, 1, 0, None)]
            sc = None if self.downsample is None else (self.downsample[0].weight, self.downsample[1], 1, self.stride, 0, None)
            return F2.res_block(x, main, sc, pair=pair)
        out = F2.conv2d(x, self.conv1.weight, bn=self.bn1, relu=True)
        if isinstance(self.conv2, ModulatedDeformConvPack):
            out = self.conv2(out, bn=self.bn2, relu=True)
        else:
            out = F2.conv2d(out, self.conv2.weight, bn=self.bn2, stride=self.stride, pad=1, relu=True, input_relu=True)
        identity = x
        if self.downsample is not None:
            identity = F2.conv2d(x, self.downsample[0].weight, bn=self.downsample[1], stride=self.stride)
        return F2.conv2d(out, self.conv3.weight, bn=self.bn3, relu=True, residual=identity, input_relu=True)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                 dcn=None, stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if not norm_eval:
            raise NotImplementedError('the MI355X path folds frozen-statistics BN (norm_eval=True), as configs/loft_foa use')
        if style != 'pytorch' or deep_stem or avg_down or plugins is not None or conv_cfg is not None:
            raise NotImplementedError('only the plain pytorch-style ResNet of configs/loft_foa is built natively')
        if in_channels != 3 or stem_channels != 64 or tuple(dilations) != (1, 1, 1, 1):
            raise NotImplementedError('stem kernel is specialised for 3->64, dilation 1')
        self.depth, self.out_indices, self.frozen_stages = depth, tuple(out_indices), frozen_stages
        self.norm_eval, self.zero_init_residual = norm_eval, zero_init_residual
        bn_grad = norm_cfg.get('requires_grad', True)
        block, stage_blocks = self.arch_settings[depth]
        self.conv1 = ConvW(3, 64, 7)
        self.bn1 = FrozenStatBN(64, requires_grad=bn_grad)
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(stage_blocks[:num_stages]):
            planes = base_channels * 2 ** i
            blocks = []
            for j in range(nb):
                stride = strides[i] if j == 0 else 1
                blocks.append(block(inplanes, planes, stride, downsample=(j == 0 and (stride != 1 or inplanes != planes * 4)),
                                    bn_grad=bn_grad, dcn=dict(dcn) if (dcn is not None and stage_with_dcn[i]) else None))
                inplanes = planes * 4
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()

    def _freeze_stages(self):
        """resnet.py:573-589.  frozen_stages < 0 (a trainable stem) is not built: the stem kernels have no backward, and its
        parameters would sit in the optimiser arena receiving weight decay with a zero gradient (configs/loft_foa use 1)."""
        if self.frozen_stages < 0:
            raise NotImplementedError('ResNet(frozen_stages=-1): the 7x7 stem has no weight-gradient kernel; use frozen_stages >= 0 '
                                      '(configs/_base_/models/bonai_loft_foa_r50_fpn_basic.py:24 sets 1)')
        if self.frozen_stages >= 0:
            for p in list(self.conv1.parameters()) + list(self.bn1.parameters()):
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, f'layer{i}').parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        """resnet.py:591-621 (pretrained checkpoints load through load_state_dict by key)."""
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
            return
        for m in self.modules():
            if isinstance(m, ConvW):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, FrozenStatBN):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():                      # resnet.py:608-612
            if isinstance(m, ModulatedDeformConvPack):
                nn.init.zeros_(m.conv_offset.weight)
                nn.init.zeros_(m.conv_offset.bias)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def forward(self, img):
        """img fp32 NCHW [B,3,H,W] -> tuple of bf16 NHWC-in-memory maps (C2..C5)."""
        with torch.no_grad():                               # frozen stem (frozen_stages >= 0, enforced in _freeze_stages)
            if getattr(self, 'compute_dtype', None) == torch.float32:   # fp32 parity mode (else: the library's 16-bit type)
                scale, shift = self.bn1.fold()
                x = K.stem7x7_bn_relu(img, self.conv1.weight, scale, shift, out_dtype=torch.float32)
            else:
                # frozen stem: BN fold + 16-bit packing once per weight version (nine small launches per step otherwise),
                # cached on the parameter like every frozen layer's operands (nn._pack_cache_*)
                bn = self.bn1
                srcs = (self.conv1.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
                key = ('stem', K.L.act16())
                packed = F2._pack_cache_get(key, srcs) if all(F2._cacheable(t) for t in srcs[:3]) else None
                if packed is None:
                    packed = K.stem7x7_pack(self.conv1.weight, *bn.fold())
                    if all(F2._cacheable(t) for t in srcs[:3]):
                        F2._pack_cache_put(key, srcs, packed)
                x = K.stem7x7_mfma(img, self.conv1.weight, None, None, packed=packed)
            x = K.maxpool3x3s2(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            layer = getattr(self, name)
            with torch.set_grad_enabled(torch.is_grad_enabled() and i + 1 > self.frozen_stages):
                # (the blocks of a stage share a hand-over dict: block k's last 1x1 and block k+1's first 1x1 run as one launch
                #  where the library serves the shape -- nn.res_block, loft_bneck_pair_bf16)
                blocks = list(layer)
                pair = {}
                for j, blk in enumerate(blocks):
                    nxt = blocks[j + 1] if j + 1 < len(blocks) else None
                    pair['next'] = ((nxt.conv1.weight, nxt.bn1) if (nxt is not None and isinstance(nxt, Bottleneck) and nxt.one_node()
                                                                     and nxt.downsample is None) else None)
                    x = blk(x, pair) if isinstance(blk, Bottleneck) else blk(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self


class _ConvModule(nn.Module):
    """`.conv.weight/.bias` naming of mmcv ConvModule (no norm, optional ReLU)."""

    def __init__(self, cin, cout, k, relu=False, conv_cfg=None):
        super().__init__()
        if conv_cfg is not None:
            self.conv = ModulatedDeformConvPack(cin, cout, k, padding=k // 2, deform_groups=_dcn_groups(conv_cfg), bias=True)
        else:
            self.conv = ConvW(cin, cout, k, bias=True)
        self.k, self.relu = k, relu

    def forward(self, x):
        if isinstance(self.conv, ModulatedDeformConvPack):
            return self.conv(x, relu=self.relu)
        return F2.conv2d(x, self.conv.weight, self.conv.bias, pad=self.k // 2, relu=self.relu)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        if add_extra_convs or norm_cfg is not None or act_cfg is not None or start_level != 0 \
                or end_level != -1 or upsample_cfg.get('mode', 'nearest') != 'nearest':
            raise NotImplementedError('only the plain FPN of configs/loft_foa (no extra convs / norm / act) is built natively')
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.num_ins = len(in_channels)
        self.lateral_convs = nn.ModuleList([_ConvModule(c, out_channels, 1, conv_cfg=conv_cfg) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_ConvModule(out_channels, out_channels, 3, conv_cfg=conv_cfg) for _ in in_channels])

    def init_weights(self):
        """fpn.py:149-154."""
        for m in self.modules():
            if isinstance(m, ConvW):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        assert len(inputs) == self.num_ins
        lats = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        lats = F2.fpn_top_down(lats)
        outs = [c(x) for c, x in zip(self.fpn_convs, lats)]
        while len(outs) < self.num_outs:           # F.max_pool2d(outs[-1], 1, stride=2)
            outs.append(F2.subsample2(outs[-1]))
        return tuple(outs)
