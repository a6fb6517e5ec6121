"""Loss modules with the reference's names and call signature
(mmdet/models/losses/cross_entropy_loss.py:128-200, smooth_l1_loss.py:52-136, utils.py:26-52,
accuracy.py:4-48).  These are O(#sampled boxes) fp32 reductions ("tiny");
round 1 evaluates them with device-side elementwise ops on the fp32 head outputs."""
import torch
import torch.nn.functional as F
from torch import nn

from .builder import LOSSES


class _FusedLossFn(torch.autograd.Function):
    """Loss value + gradient in ONE launch (loft_fused_loss_v2); backward is the stored gradient times the incoming scalar."""

    @staticmethod
    def forward(ctx, pred, mode, target, weight, avg_factor, count, scale, beta, out_shape, want_acc, target_ge1):
        from .. import kernels as K
        res = K.fused_loss(mode, pred, target, weight, avg_factor, count, scale, beta, want_acc=want_acc, target_ge1=target_ge1)
        ctx.save_for_backward(res[1])
        ctx.meta = (tuple(pred.shape), pred.dtype)
        if want_acc:
            ctx.mark_non_differentiable(res[2])
            return res[0].reshape(out_shape), res[2]
        return res[0].reshape(out_shape)

    @staticmethod
    def backward(ctx, g, *unused):
        (grad,) = ctx.saved_tensors
        shape, dt = ctx.meta
        gp = (grad * g.reshape(())).reshape(shape)
        return (gp if dt == torch.float32 else gp.to(dt)), None, None, None, None, None, None, None, None, None, None


ELEMENTWISE_ONLY = False   # tests: compare the fused value+gradient launch with the elementwise formulation below


def _fusable(pred, reduction):
    """The product path: device tensors, the configs' 'mean' reduction.  Host tensors (CPU-side tests) and the other reductions
    keep the elementwise formulation below."""
    return pred.is_cuda and reduction == 'mean' and pred.numel() > 0 and not ELEMENTWISE_ONLY


def _fused(mode, pred, target, weight, avg_factor, scale, beta=1.0, count=None, out_shape=(), want_acc=False, target_ge1=False):
    return _FusedLossFn.apply(pred, mode, target, weight, avg_factor, count, float(scale), float(beta), out_shape, want_acc,
                              target_ge1)


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
        if self.class_weight is None and not kwargs and _fusable(cls_score, reduction):
            if self.use_sigmoid:
                if cls_score.dim() != label.dim():          # RPN: labels in {0,1} -> one channel
                    if cls_score.numel() == label.numel():     # (label >= 1 is evaluated by the loss kernel on the int64 labels)
                        return _fused('bce', cls_score.reshape(-1), label.reshape(-1), None if weight is None else
                                      weight.reshape(-1), avg_factor, self.loss_weight, target_ge1=True)
                else:
                    return _fused('bce', cls_score, label, weight, avg_factor, self.loss_weight)
            elif self.use_mask:
                # (here the third positional argument carries the RoIs' class labels, fcn_mask_head.py:143-149; with one mask
                #  channel they select nothing)
                if cls_score.shape[1] == 1 and avg_factor is None:
                    return _fused('bce', cls_score.reshape(-1), label.reshape(-1), None, None, self.loss_weight, out_shape=(1,))
            elif cls_score.dim() == 2:
                # the same launch counts the top-1 hits: `accuracy(cls_score, label)` of the caller (bbox_head.py:152) for free
                loss, acc = _fused('ce', cls_score, label, weight, avg_factor, self.loss_weight, want_acc=True)
                self.last_accuracy = (cls_score, label, acc)
                return loss
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
        if _fusable(pred, reduction):
            return _fused('l1', pred, target, weight, avg_factor, self.loss_weight)
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if not kwargs and _fusable(pred, reduction):
            return _fused('smooth_l1', pred, target, weight, avg_factor, self.loss_weight, beta=self.beta)
        d = (pred - target).abs()
        loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


def accuracy(pred, target, topk=1, loss_module=None):
    """accuracy.py:4-48 (top-1, percent).  loss_module: a CrossEntropyLoss that has just been evaluated on this very
    (pred, target) pair -- its fused launch already counted the hits."""
    if pred.size(0) == 0:
        return pred.new_zeros(1)
    last = getattr(loss_module, 'last_accuracy', None)
    if last is not None and last[0] is pred and last[1] is target:
        loss_module.last_accuracy = None
        return last[2]
    correct = (pred.argmax(dim=1) == target).float().sum()
    return (correct * (100.0 / pred.size(0))).view(1)


@LOSSES.register_module()
class MSELoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        return self.loss_weight * weight_reduce_loss((pred - target) ** 2, weight, self.reduction, avg_factor)
