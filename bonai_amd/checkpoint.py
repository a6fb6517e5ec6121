"""Checkpoint I/O in the reference's formats.

``load_checkpoint`` accepts what mmcv.runner.load_checkpoint accepts for this model family (call sites
mmdet/models/backbones/resnet.py:591-600, mmdet/apis/train.py:139-142, tools/test.py): a path or an already loaded
object that is either a bare ``state_dict`` or ``{'meta': ..., 'state_dict': ..., 'optimizer': ...}``; keys may carry
the ``module.`` prefix of (MM)DataParallel wrappers; a torchvision ResNet checkpoint (``conv1.weight``, ``layer1.0...``,
``fc.*``) loads into ``model.backbone``.  ``save_checkpoint`` writes the reference layout (meta / state_dict / optimizer,
tensors on CPU) so tools/publish_model.py-style consumers read it back.
"""
from collections import OrderedDict

import torch


def _strip(sd, prefix):
    return OrderedDict((k[len(prefix):] if k.startswith(prefix) else k, v) for k, v in sd.items())


def normalise_state_dict(obj):
    """-> OrderedDict of tensors with ``module.`` prefixes removed."""
    if isinstance(obj, dict) and 'state_dict' in obj and isinstance(obj['state_dict'], dict):
        obj = obj['state_dict']
    if not isinstance(obj, dict):
        raise RuntimeError(f'no state_dict found in checkpoint of type {type(obj)}')
    sd = OrderedDict(obj)
    while any(k.startswith('module.') for k in sd):
        sd = _strip(sd, 'module.')
    return sd


def load_checkpoint(model, checkpoint, map_location='cpu', strict=False, logger=None):
    """Load into ``model`` (a detector, or a backbone).  Returns the loaded checkpoint dict (like mmcv)."""
    ckpt = torch.load(checkpoint, map_location=map_location) if isinstance(checkpoint, str) else checkpoint
    sd = normalise_state_dict(ckpt)
    own = model.state_dict()
    if not any(k in own for k in sd):
        # backbone-only checkpoint (torchvision / open-mmlab model zoo) offered to a detector, or the reverse
        if any(('backbone.' + k) in own for k in sd):
            sd = OrderedDict(('backbone.' + k, v) for k, v in sd.items())
        elif any(k.startswith('backbone.') and k[len('backbone.'):] in own for k in sd):
            sd = OrderedDict((k[len('backbone.'):], v) for k, v in sd.items() if k.startswith('backbone.'))
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    bad = [k for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError('size mismatch for ' + ', '.join(f'{k}: {tuple(sd[k].shape)} vs {tuple(own[k].shape)}' for k in bad[:8]))
    if strict and (missing or unexpected):
        raise RuntimeError(f'missing keys {missing[:8]}..., unexpected keys {unexpected[:8]}...')
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    if logger is not None and (missing or unexpected):
        logger.warning(f'load_checkpoint: {len(missing)} missing, {len(unexpected)} unexpected keys '
                       f'(e.g. {missing[:3]} / {unexpected[:3]})')
    return ckpt if isinstance(ckpt, dict) else {'state_dict': sd}


def save_checkpoint(model, filename, optimizer_state=None, meta=None):
    """{'meta', 'state_dict', 'optimizer'} with CPU tensors (mmcv.runner.save_checkpoint layout)."""
    if hasattr(model, 'module'):
        model = model.module
    ckpt = dict(meta=dict(meta or {}), state_dict=OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items()))
    if optimizer_state is not None:
        ckpt['optimizer'] = optimizer_state
    torch.save(ckpt, filename)
    return filename
