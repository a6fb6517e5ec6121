"""Loss modules with the reference's names and call signature
(mmdet/models/losses/cross_entropy_loss.py:128-200, smooth_l1_loss.py:52-136, utils.py:26-52,
accuracy.py:4-48).  These are O(#sampled boxes) fp32 reductions ("tiny");
round 1 evaluates them with device-side elementwise ops on the fp32 head outputs."""
import torch
import torch.nn.functional as F
from torch import nn

from .builder import LOSSES


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    if pred.dim() != label.dim():  # RPN: labels in {0,1} -> one channel
        tgt = (label >= 1).float().view(-1, 1)
        if weight is not None:
            weight = weight.view(-1, 1).expand(weight.size(0), pred.size(-1))
    else:
        tgt = label.float()
    loss = F.binary_cross_entropy_with_logits(pred, tgt, weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


def mask_cross_entropy(pred, target, label, reduction='mean', avg_factor=None, class_weight=None):
    assert reduction == 'mean' and avg_factor is None
    inds = torch.arange(pred.size(0), dtype=torch.long, device=pred.device)
    return F.binary_cross_entropy_with_logits(pred[inds, label], target, weight=class_weight, reduction='mean')[None]


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not (use_sigmoid and use_mask)
        self.use_sigmoid, self.use_mask, self.reduction = use_sigmoid, use_mask, reduction
        self.loss_weight, self.class_weight = loss_weight, class_weight
        self.cls_criterion = binary_cross_entropy if use_sigmoid else (mask_cross_entropy if use_mask else cross_entropy)

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        cw = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=cw, reduction=reduction,
                                                     avg_factor=avg_factor, **kwargs)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        d = (pred - target).abs()
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


def accuracy(pred, target, topk=1):
    if pred.size(0) == 0:
        return pred.new_zeros(1)
    correct = (pred.argmax(dim=1) == target).float().sum()
    return (correct * (100.0 / pred.size(0))).view(1)


@LOSSES.register_module()
class MSELoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        return self.loss_weight * weight_reduce_loss((pred - target) ** 2, weight, self.reduction, avg_factor)
