"""HRNet backbone + HRFPN neck on the MI355X kernels (BASELINE config 5).

Mirrors the constructor surface, module tree and parameter names of the reference's ``HRNet`` / ``HRModule``
(mmdet/models/backbones/hrnet.py:12-537), ``BasicBlock`` (mmdet/models/backbones/resnet.py:13-92) and ``HRFPN``
(mmdet/models/necks/hrfpn.py:11-102) so ``configs/hrnet/*`` style ``extra`` dicts build unchanged and reference /
``open-mmlab://msra/hrnetv2_w32`` checkpoints load by key.

How it runs here: NHWC bf16 activations; every conv+BN(+ReLU)(+residual) is one MFMA tap-conv launch with the
frozen-statistics BN folded into the weights (norm_eval=True, hrnet.py:527-537); the 32-channel branch is carried in
64-channel tensors whose upper half is exactly zero (zero-padded weight packings -- the MFMA K-step is 64 channels), the
fuse step ``relu(sum_j f_ij(x_j))`` is one kernel that reads every term at its own resolution (loft_fuse_sum_relu), the
3-channel stem conv is a dedicated kernel, and HRFPN's bilinear upsample + concat / avg-pool pyramid are
loft_bilinear_up_slot / loft_avgpool.
"""
import torch
from torch import nn

from .. import kernels as K
from .. import nn as F2
from ..debug import DBG
from .backbone import Bottleneck, ConvW, FrozenStatBN
from .builder import BACKBONES, NECKS

PAD = 64      # narrowest channel count carried in memory
BRANCH_STREAMS = None     # list of >= 3 torch.cuda.Stream: HRModule branches 1.. on their own streams (see HRModule._run_branches)


def _p(c):
    return max(c, PAD)


class _Seq(nn.Sequential):
    """conv (+ bn) (+ relu) under the reference's Sequential indices ('0' = conv, '1' = bn)."""

    def __init__(self, cin, cout, k, stride=1, relu=False):
        super().__init__()
        self.add_module('0', ConvW(cin, cout, k))
        self.add_module('1', FrozenStatBN(cout))
        self.k, self.stride, self.relu, self.cout = k, stride, relu, cout

    def forward(self, x, residual=None):
        return F2.conv2d(x, self[0].weight, bn=self[1], stride=self.stride, pad=self.k // 2, relu=self.relu,
                         residual=residual, cout_pad=_p(self.cout))


class BasicBlock(nn.Module):
    """resnet.py:13-92: conv3x3 - bn - relu - conv3x3 - bn, += identity, relu."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvW(inplanes, planes, 3)
        self.bn1 = FrozenStatBN(planes)
        self.conv2 = ConvW(planes, planes, 3)
        self.bn2 = FrozenStatBN(planes)
        self.stride, self.planes = stride, planes
        self.downsample = downsample

    def forward(self, x):
        if not DBG.no_block_fusion:     # one autograd node per block: no aten add / separate ReLU-backward per block
            cp = _p(self.planes)
            main = [(self.conv1.weight, self.bn1, 3, self.stride, 1, cp), (self.conv2.weight, self.bn2, 3, 1, 1, cp)]
            ds = self.downsample
            sc = None if ds is None else (ds[0].weight, ds[1], ds.k, ds.stride, ds.k // 2, _p(ds.cout))
            return F2.res_block(x, main, sc)
        out = F2.conv2d(x, self.conv1.weight, bn=self.bn1, stride=self.stride, pad=1, relu=True, cout_pad=_p(self.planes))
        identity = x if self.downsample is None else self.downsample(x)
        return F2.conv2d(out, self.conv2.weight, bn=self.bn2, pad=1, relu=True, residual=identity, cout_pad=_p(self.planes))


class _Bottleneck(Bottleneck):
    """ResNet Bottleneck with the HRNet-style ``downsample`` Sequential (conv '0', bn '1')."""

    def __init__(self, inplanes, planes, downsample):
        super().__init__(inplanes, planes, stride=1, downsample=downsample)


BLOCKS = {'BASIC': BasicBlock, 'BOTTLENECK': _Bottleneck}


def _make_layer(block, inplanes, planes, blocks):
    """hrnet.py:393-424 (_make_layer) / :62-103 (_make_one_branch), stride 1."""
    layers = []
    for i in range(blocks):
        need_ds = i == 0 and inplanes != planes * block.expansion
        if block is BasicBlock:
            ds = _Seq(inplanes, planes, 1) if need_ds else None
            layers.append(BasicBlock(inplanes, planes, downsample=ds))
        else:
            layers.append(_Bottleneck(inplanes, planes, need_ds))
        inplanes = planes * block.expansion
    return nn.Sequential(*layers)


class HRModule(nn.Module):
    """hrnet.py:12-195."""

    def __init__(self, num_branches, block, num_blocks, in_channels, num_channels, multiscale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(in_channels)):
            raise ValueError('HRModule: branch / block / channel counts disagree')
        self.num_branches = num_branches
        self.in_channels = list(in_channels)
        self.branches = nn.ModuleList()
        for i in range(num_branches):
            self.branches.append(_make_layer(block, self.in_channels[i], num_channels[i], num_blocks[i]))
            self.in_channels[i] = num_channels[i] * block.expansion
        self.multiscale_output = multiscale_output
        self.fuse_layers = self._make_fuse_layers()

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.in_channels
        fuse = []
        for i in range(nb if self.multiscale_output else 1):
            row = []
            for j in range(nb):
                if j > i:       # 1x1 conv + bn, nearest-upsampled by 2^(j-i) inside the fuse kernel (hrnet.py:130-143)
                    row.append(_Seq(ch[j], ch[i], 1))
                elif j == i:
                    row.append(None)
                else:           # chain of 3x3 stride-2 convs (hrnet.py:146-172)
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        chain.append(_Seq(ch[j], ch[i] if last else ch[j], 3, stride=2, relu=not last))
                    row.append(nn.Sequential(*chain))
            fuse.append(nn.ModuleList(row))
        return nn.ModuleList(fuse)

    def _run_branches(self, x):
        """The branches of a module are independent chains of small launches (hrnet.py:177-183 runs them one after the other):
        with BRANCH_STREAMS set (hipGraph capture: bonai_amd/graphs.py; tests) branch i > 0 runs on its own HIP stream, forked
        from and joined to the calling stream -- in a captured graph that makes them parallel branches, and autograd replays the
        backward on the same streams."""
        streams = BRANCH_STREAMS
        if not streams or not x[0].is_cuda or len(streams) < self.num_branches - 1:
            return [self.branches[i](x[i]) for i in range(self.num_branches)]
        main = torch.cuda.current_stream()
        ys = [None] * self.num_branches
        for i in range(1, self.num_branches):
            s = streams[i - 1]
            s.wait_stream(main)
            x[i].record_stream(s)
            with torch.cuda.stream(s):
                ys[i] = self.branches[i](x[i])
        ys[0] = self.branches[0](x[0])
        for i in range(1, self.num_branches):
            main.wait_stream(streams[i - 1])
            ys[i].record_stream(main)
        return ys

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        x = self._run_branches(list(x))
        out = []
        for i in range(len(self.fuse_layers)):
            terms, shifts = [], []
            for j in range(self.num_branches):
                if j == i:
                    terms.append(x[j]); shifts.append(0)
                elif j > i:
                    terms.append(self.fuse_layers[i][j](x[j])); shifts.append(j - i)
                else:
                    t = x[j]
                    for m in self.fuse_layers[i][j]:
                        t = m(t)
                    terms.append(t); shifts.append(0)
            out.append(F2.fuse_sum_relu(terms, shifts))
        return out


@BACKBONES.register_module()
class HRNet(nn.Module):
    """hrnet.py:198-537.  Returns the branch maps; the 32-channel one is 64 channels wide in memory (upper half zero)."""

    def __init__(self, extra, in_channels=3, conv_cfg=None, norm_cfg=dict(type='BN'), norm_eval=True, with_cp=False,
                 zero_init_residual=False):
        super().__init__()
        if in_channels != 3 or conv_cfg is not None or not norm_eval or norm_cfg.get('type', 'BN') != 'BN':
            raise NotImplementedError('native HRNet: 3-channel input, plain conv, frozen-statistics BN (norm_eval=True)')
        self.extra, self.zero_init_residual = extra, zero_init_residual
        self.compute_dtype = None       # torch.float32: fp32 parity mode; anything else: the library's 16-bit type
        self.conv1 = ConvW(3, 64, 3)
        self.bn1 = FrozenStatBN(64)
        self.conv2 = ConvW(64, 64, 3)
        self.bn2 = FrozenStatBN(64)
        s1 = extra['stage1']
        block = BLOCKS[s1['block']]
        self.layer1 = _make_layer(block, 64, s1['num_channels'][0], s1['num_blocks'][0])
        pre = [s1['num_channels'][0] * block.expansion]
        for si in (2, 3, 4):
            cfg = extra[f'stage{si}']
            block = BLOCKS[cfg['block']]
            chans = [c * block.expansion for c in cfg['num_channels']]
            setattr(self, f'transition{si - 1}', self._make_transition(pre, chans))
            mods, cur = [], list(chans)
            for _ in range(cfg['num_modules']):
                m = HRModule(cfg['num_branches'], block, cfg['num_blocks'], cur, cfg['num_channels'])
                cur = m.in_channels
                mods.append(m)
            setattr(self, f'stage{si}', nn.Sequential(*mods))
            pre = cur
        self.out_channels = pre

    @staticmethod
    def _make_transition(pre, cur):
        """hrnet.py:345-391."""
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(_Seq(pre[i], cur[i], 3, relu=True) if cur[i] != pre[i] else None)
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    chain.append(_Seq(pre[-1], cout, 3, stride=2, relu=True))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    def init_weights(self, pretrained=None):
        """hrnet.py:461-484."""
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
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    @staticmethod
    def _run(t, layer):
        if isinstance(layer, nn.Sequential) and not isinstance(layer, _Seq):
            for m in layer:
                t = m(t)
            return t
        return layer(t)

    def forward(self, img):
        x = F2.stem3x3s2(img, self.conv1.weight, self.bn1, torch.float32 if self.compute_dtype == torch.float32 else K.L.act16())
        x = F2.conv2d(x, self.conv2.weight, bn=self.bn2, stride=2, pad=1, relu=True)
        x = self.layer1(x)
        y = [x]
        for si in (2, 3, 4):                            # hrnet.py:486-515
            trans = getattr(self, f'transition{si - 1}')
            src = y[-1]                                 # new / changed branches always start from the last (coarsest) map
            xs = [self._run(src, trans[i]) if trans[i] is not None else y[i]
                  for i in range(self.extra[f'stage{si}']['num_branches'])]
            for m in getattr(self, f'stage{si}'):
                xs = m(xs)
            y = xs
        return tuple(y)


class _NeckConv(nn.Module):
    """mmcv ConvModule naming (``.conv.weight/.bias``), no norm, no activation."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = ConvW(cin, cout, k, bias=True)
        self.k = k


@NECKS.register_module()
class HRFPN(nn.Module):
    """hrfpn.py:11-102."""

    def __init__(self, in_channels, out_channels, num_outs=5, pooling_type='AVG', conv_cfg=None, norm_cfg=None, with_cp=False,
                 stride=1):
        super().__init__()
        if pooling_type != 'AVG' or conv_cfg is not None or norm_cfg is not None or stride != 1:
            raise NotImplementedError('native HRFPN: AVG pooling, plain convs, stride 1 (the configs/hrnet defaults)')
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.num_ins = len(in_channels)
        self.reduction_conv = _NeckConv(sum(in_channels), out_channels, 1)
        self.fpn_convs = nn.ModuleList([_NeckConv(out_channels, out_channels, 3) for _ in range(num_outs)])

    def init_weights(self):
        """hrfpn.py:72-76: caffe2_xavier_init = kaiming_uniform(a=1, fan_in), zero bias."""
        for m in self.modules():
            if isinstance(m, ConvW):
                nn.init.kaiming_uniform_(m.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
                nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        assert len(inputs) == self.num_ins
        cat = F2.hrfpn_concat(list(inputs))            # branch i bilinearly upsampled by 2^i into its channel slot
        w = self.reduction_conv.conv.weight
        widths = [x.shape[1] for x in inputs]
        if widths != self.in_channels:                 # padded branches: spread the weight columns over the padded slots
            cols, off = [], 0
            for c, wd in zip(self.in_channels, widths):
                cols.append(w[:, off:off + c])
                if wd > c:
                    cols.append(w.new_zeros(w.shape[0], wd - c, 1, 1))
                off += c
            w = torch.cat(cols, 1)
        out = F2.conv2d(cat, w, self.reduction_conv.conv.bias)
        outs = [out] + [F2.avgpool(out, i) for i in range(1, self.num_outs)]
        return tuple(F2.conv2d(o, c.conv.weight, c.conv.bias, pad=1) for o, c in zip(outs, self.fpn_convs))
