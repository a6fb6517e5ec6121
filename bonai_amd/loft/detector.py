"""LOFT detector: the reference's two-stage orchestration on the MI355X-native pieces.

Mirrors mmdet/models/detectors/loft.py:11-32, two_stage.py:17-199 and base.py:123-243: same constructor
(backbone, neck, rpn_head, roi_head, train_cfg, test_cfg, pretrained), same ``forward(img, img_metas,
return_loss=True, **kw)`` dispatch, ``train_step`` return dict and ``log_vars`` key set.  The eight
per-key blocking all-reduces + ``.item()`` of base.py:201-206 become ONE 8-float all-reduce and one
device->host copy.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch import nn

from .. import nn as F2

from .builder import DETECTORS, build_backbone, build_head, build_neck


def resolve_pretrained(pretrained):
    """``pretrained`` of a detector config -> a local checkpoint path, or None.

    The reference hands ``torchvision://resnet50`` / ``open-mmlab://...`` to mmcv's load_checkpoint, which downloads it
    (resnet.py:591-600).  There is no network in this deployment: the URI's model name is looked up as
    ``$LOFT_PRETRAINED_DIR/<name>.pth`` (or ``.pt`` / the bare name; torchvision's ``resnet50-19c8e357.pth`` style names match by
    prefix).  When nothing is found the run continues from random initialisation -- with ``frozen_stages=1`` that means a frozen
    RANDOM stem and layer1, so it is reported with a RuntimeWarning (and an error when LOFT_PRETRAINED_STRICT=1), never silently."""
    import os
    import warnings
    if not isinstance(pretrained, str) or '://' not in pretrained:
        return pretrained
    name = pretrained.split('://', 1)[1].strip('/')
    root = os.environ.get('LOFT_PRETRAINED_DIR')
    if root and os.path.isdir(root):
        for cand in (name, name + '.pth', name + '.pt'):
            if os.path.isfile(os.path.join(root, cand)):
                return os.path.join(root, cand)
        for f in sorted(os.listdir(root)):
            if f.startswith(os.path.basename(name) + '-') and f.endswith(('.pth', '.pt')):
                return os.path.join(root, f)
    msg = (f'pretrained={pretrained!r} cannot be downloaded (no network) and no local copy was found under LOFT_PRETRAINED_DIR='
           f'{root!r}: the backbone keeps its RANDOM initialisation (frozen_stages then freezes random features). '
           'Pass a local checkpoint path as `pretrained`, set LOFT_PRETRAINED_DIR, or use tools/train.py --pretrained.')
    if os.environ.get('LOFT_PRETRAINED_STRICT') == '1':
        raise FileNotFoundError(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return None


@DETECTORS.register_module()
class LOFT(nn.Module):
    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        if rpn_head is not None:
            rpn_cfg = dict(rpn_head)
            rpn_cfg.update(train_cfg=train_cfg.rpn if train_cfg is not None else None, test_cfg=test_cfg.rpn)
            self.rpn_head = build_head(rpn_cfg)
        if roi_head is not None:
            roi_cfg = dict(roi_head)
            roi_cfg.update(train_cfg=train_cfg.rcnn if train_cfg is not None else None, test_cfg=test_cfg.rcnn)
            self.roi_head = build_head(roi_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.with_vis_feat = True
        self.init_weights(pretrained)

    with_neck = property(lambda self: self.neck is not None)
    with_rpn = property(lambda self: hasattr(self, 'rpn_head'))
    with_roi_head = property(lambda self: hasattr(self, 'roi_head'))

    def init_weights(self, pretrained=None):
        """two_stage.py:60-78.  Model-zoo URIs (``torchvision://resnet50``, bonai_loft_foa_r50_fpn_basic.py:4) resolve to a local
        file (no network here) or warn loudly: see ``resolve_pretrained``."""
        pretrained = resolve_pretrained(pretrained)
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_rpn:
            self.rpn_head.init_weights()
        if self.with_roi_head:
            self.roi_head.init_weights(pretrained)

    feat_provider = None      # the running Trainer's bonai_amd.graphs.FeatureGraphs.provider (hipGraph replay of this function)

    def extract_feat(self, img):
        if self.feat_provider is not None and torch.is_grad_enabled():
            return self.feat_provider(img)
        x = self.backbone(img)
        # mixed precision (a measurement mode, bench.py `value_mixed`): the 16-bit training kernels up to a boundary, the fp32
        # parity mode's arithmetic (fp32 activations, operand-plane contractions) behind it.  'neck': backbone trunk 16-bit,
        # FPN + RPN + RoI heads fp32; 'heads': backbone + FPN 16-bit, RPN + RoI heads fp32; 'trunk': backbone + FPN fp32-grade,
        # RPN + RoI heads 16-bit.  The cast is an autograd op.
        mixed = getattr(self, 'mixed_precision', None)
        if mixed == 'neck':
            x = tuple(f.float() for f in x)
        if self.with_neck:
            x = self.neck(x)
        if mixed == 'heads':
            x = tuple(f.float() for f in x)
        if mixed == 'trunk':
            # the reverse split: backbone + FPN in the fp32 parity mode (the caller sets
            # backbone.compute_dtype = torch.float32), RPN + RoI heads on the 16-bit kernels
            x = tuple(f.to(F2.K.L.act16()) for f in x)
        return x

    def forward_dummy(self, img):
        """two_stage.py:87-103 (used by tools/get_flops.py): backbone + neck + RPN + RoI heads on 1000 random proposals."""
        outs = ()
        x = self.extract_feat(img)
        if self.with_rpn:
            outs = outs + (self.rpn_head(x),)
        proposals = torch.randn(1000, 4, device=img.device)
        return outs + (self.roi_head.forward_dummy(x, proposals),)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      gt_offsets=None, **kwargs):
        if self.with_rpn and proposals is None and hasattr(self.rpn_head, 'prefetch_targets'):
            self.rpn_head.prefetch_targets(img, gt_bboxes)      # anchor assignment + sampling beside the backbone (side stream)
        x = self.extract_feat(img)
        if F2.HUB_ENABLED and x[0].is_cuda and x[0].dtype == F2.K.L.act16():
            # one shared gradient map per pyramid level for the RPN head and the three RoI extractors (nn.feat_hub)
            x = F2.feat_hub(x, 4)
        losses = dict()
        if self.with_rpn:
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
            rpn_losses, proposal_list = self.rpn_head.forward_train(x, img_metas, gt_bboxes, gt_labels=None,
                                                                    gt_bboxes_ignore=gt_bboxes_ignore,
                                                                    proposal_cfg=proposal_cfg)
            losses.update(rpn_losses)
        else:
            proposal_list = proposals
        losses.update(self.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore,
                                                  gt_masks, gt_offsets=gt_offsets, **kwargs))
        return losses

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        if len(imgs) != 1:
            raise NotImplementedError('test-time augmentation (configs/loft_foa: flip=False, one scale)')
        assert imgs[0].size(0) == 1, 'samples_per_gpu must be 1 at test time (base.py:142-143)'
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        x = self.extract_feat(img)
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)

    @staticmethod
    def _parse_losses(losses):
        """base.py:175-208, with the per-key all_reduce + .item() folded into one collective / one copy."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean() if value.numel() != 1 else value.reshape(())     # (the mean of one element: no launch)
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() if v.numel() != 1 else v.reshape(()) for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        terms = [v for k, v in log_vars.items() if 'loss' in k]
        if len(terms) > 2 and all(t.is_cuda for t in terms):
            # one stack + one reduction instead of a chain of len - 1 additions at the very end of the forward pass (each a 5 us
            # launch the backward pass waits for); the logged vector below is ONE more stack: 'loss' rides in it as the last term
            loss = torch.stack([t.float().reshape(()) for t in terms]).sum()
        else:
            loss = sum(terms)
        log_vars['loss'] = loss
        vec = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # asynchronous: the collective runs on the process group's own stream behind the forward pass and is waited for
            # only when somebody reads the values (_LazyLogVars) -- issued synchronously it would make every rank's main stream
            # wait for the slowest rank between forward and backward.  Same issue order on every rank (before the gradient
            # buckets of this step), as RCCL requires.
            work = dist.all_reduce(vec, async_op=True)
            vec = (vec, work, float(dist.get_world_size()))
        return loss, log_vars, vec

    def train_step(self, data, optimizer=None):
        from .. import nn as F2
        F2._USES.clear()                                    # consumer counts of this step's ReLU outputs (nn._note_use)
        losses = self(**data)
        loss, log_vars, vec = self._parse_losses(losses)
        return dict(loss=loss, log_vars=_LazyLogVars(list(log_vars.keys()), vec), num_samples=len(data['img_metas']))


class _LazyLogVars(OrderedDict):
    """log_vars whose float values are fetched from the device only when read (no per-step host sync)."""

    def __init__(self, keys, vec):
        super().__init__((k, None) for k in keys)
        self._vec, self._done = vec, False

    def _materialise(self):
        if not self._done:
            if isinstance(self._vec, tuple):            # (sum over ranks in flight, its work handle, world size)
                vec, work, world = self._vec
                work.wait()
                self._vec = vec / world
            vals = self._vec.tolist()
            for k, v in zip(list(super().keys()), vals):
                super().__setitem__(k, v)
            self._done = True

    def __getitem__(self, k):
        self._materialise()
        return super().__getitem__(k)

    def items(self):
        self._materialise()
        return super().items()

    def values(self):
        self._materialise()
        return super().values()
