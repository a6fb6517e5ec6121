"""Deterministic, name-keyed synthetic weights for parity runs.  TEST INFRASTRUCTURE.

81 M parameters cannot be committed as a fixture, so both sides of every end-to-end comparison
(the reference in the build container, the oracle and the HIP model on the GPU box) regenerate the same
tensors from the parameter NAME: generator seed = crc32(name), scale by role.  torch's CPU generator is
bit-reproducible for a fixed torch build (the GPU box runs the same image)."""
import zlib

import torch


def _gen(name):
    return torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)


def synth_tensor(name, shape, dtype=torch.float32):
    g = _gen(name)
    shape = tuple(shape)
    if name.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith('running_var'):
        return torch.rand(shape, generator=g) * 1.0 + 0.5
    if name.endswith('running_mean'):
        return torch.randn(shape, generator=g) * 0.1
    leaf = name.rsplit('.', 2)
    is_bn = len(leaf) >= 2 and (leaf[-2].startswith('bn') or (leaf[-2] == '1' and 'downsample' in name))
    hr_seq_bn = len(shape) == 1 and ('fuse_layers' in name or 'transition' in name) and leaf[-2] == '1'   # HRNet Sequential(conv, bn)
    if hr_seq_bn and name.endswith('weight'):
        return torch.rand(shape, generator=g) * 0.3 + 0.3
    if is_bn and name.endswith('weight'):
        if leaf[-2] == 'bn3' or (leaf[-2] == 'bn2' and 'branches' in name):   # residual branch: keep the stacked sums well conditioned
            return torch.rand(shape, generator=g) * 0.2 + 0.2
        return torch.rand(shape, generator=g) * 0.4 + 0.8
    if name.endswith('bias'):
        return torch.randn(shape, generator=g) * 0.02
    final = any(t in name for t in ('rpn_cls', 'rpn_reg', 'fc_cls', 'fc_reg', 'fc_offset', 'conv_logits'))
    gain = 0.01 if final else (1.0 if ('lateral' in name or 'fpn_convs' in name) else 2.0)
    if 'conv_offset' in name:   # DCNv2 offsets of ~0.3 px: keeps 20 stacked data-dependent samplers well conditioned
        gain = 0.1
    if len(shape) == 4:       # conv / deconv weight
        fan_in = shape[1] * shape[2] * shape[3]
        if 'upsample' in name:
            fan_in = shape[0]
        return torch.randn(shape, generator=g) * (gain / fan_in) ** 0.5
    if len(shape) == 2:       # linear
        return torch.randn(shape, generator=g) * (gain / shape[1]) ** 0.5
    return torch.randn(shape, generator=g) * 0.1


def synth_state_dict(reference_state_dict):
    """Same keys/shapes as the given state_dict, deterministic values."""
    return {k: synth_tensor(k, v.shape) for k, v in reference_state_dict.items()}
