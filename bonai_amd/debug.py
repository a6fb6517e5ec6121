"""The A/B switches of the host-side plumbing, in ONE place.

Every switch turns OFF one fusion / stream / pooling decision of the trainer and selects the formulation it replaced; none is a
fallback for a missing extension (bonai_amd.lib raises when the library does not load) and the kernels never read the
environment.  The product runs with all of them False.  They exist for same-box A/B timing (DESIGN.md "Same-box A/B switches"),
for the serialised mode of the roofline instrumentation, and for tests/test_trainer_gpu.py, which checks that each switch alone
reproduces the default training step.

Set them in-process::

    from bonai_amd.debug import DBG
    with DBG.override(no_side_stream=True):
        ...

or from the shell, read ONCE when this module is imported: ``LOFT_NO_SIDE_STREAM=1 python bench.py`` (the environment name is
``LOFT_`` + the upper-cased field name)."""
import contextlib
import os

SWITCHES = {
    # streams
    'no_side_stream': 'mask / bbox branches, the RPN proposal chain and the backbone weight gradients all on the main stream '
                      '(the mode of bench.py\'s roofline instrumentation)',
    'no_bbox_side_stream': 'the bbox head on the main stream (the mask branch keeps its side stream)',
    'no_rpn_side_stream': 'the RPN proposal chain (sort, decode, NMS) on the main stream',
    'no_rpn_target_prefetch': 'RPN anchor assignment + sampling after the RPN convs on the main stream, not beside the backbone',
    'no_rpn_loss_stream': 'the RPN losses (and with them the sparse RPN backward) on the main stream',
    'no_wgrad_stream': 'backbone weight-gradient launches on the data-gradient stream',
    'unpack_stream': 'the batched weight-gradient unpack launches (descriptor upload + loft_fold_unpack_bwd_multi) on a stream of their '
                     'own instead of on the main stream between the data-gradient kernels (round 6: measured, not the default)',
    # batched launches / pools of the trainer
    'no_prepack': 'BN fold + operand packing per conv per step instead of one launch per step (kernels.PrepackRegistry)',
    'no_unpack_queue': 'one loft_fold_unpack_bwd launch per conv instead of the batched unpack (kernels.UnpackQueue)',
    'no_grad_sink': 'gradients returned to autograd and accumulated by it, not deposited in the arena by the kernels',
    'no_leaf_sink': 'narrow heads and the sparse RPN backward return their parameter gradients to autograd (per-parameter accumulation '
                    'launches) instead of depositing them through the unpack queue (nn._queue_param_grads)',
    'no_head_fusion': 'narrow 1x1 heads (RPN objectness + deltas, mask logits) as launches of their own instead of in the producing '
                      'conv\'s epilogue (loft_conv_tap_bf16_head)',
    'no_grad_join': 'autograd sums the two data gradients of a backbone stage output (next stage + FPN lateral) with an elementwise add '
                    'instead of the residual block taking the lateral\'s deposit as its data-gradient residual (nn.JOIN)',
    'no_deconv_fusion': 'the mask head\'s 2x2 deconvolution as four parity launches instead of one launch whose four channel tiles are '
                        'the four taps (loft_deconv2x2_bf16: the input tile read once)',
    'no_zero_pool': 'torch.zeros / torch.empty per accumulation buffer instead of the step\'s pre-zeroed / scratch slabs',
    'no_feat_hub': 'autograd sums the RPN / RoI-extractor gradients of the FPN maps (no shared per-level gradient map)',
    # autograd-node granularity / previous formulations of three backward ops
    'no_block_fusion': 'one autograd node per conv of a residual block instead of nn.res_block',
    'no_linear_fn': 'Linear layers through the generic conv node',
    'no_bneck_fusion': 'the frozen / inference 64-plane bottleneck as three (four) tap-conv launches instead of conv1 + the fused tail '
                       '(loft_bneck_tail_bf16: 3x3 + 1x1 expansion + shortcut + ReLU in one launch)',
    'no_pair_fusion': 'the last 1x1 of bottleneck k and the first 1x1 of bottleneck k+1 (layer2 / layer3) as two launches each way instead '
                      'of one (loft_bneck_pair_bf16: the block output is written once and not re-read as the next conv1\'s operand)',
    'wgrad_no_patch': 'the tap weight-gradient kernel for 64-channel high-resolution layers (no patch kernel)',
    'roi_fp32_bwd': 'RoIAlign backward into fp32 maps + a cast',
    'roi_sort': 'RoIAlign forward workgroups in (image, level, row strip) launch order for lists of >= 256 RoIs (loft_roi_order; '
                'measured neutral on the step, not shipped)',
    'no_roi_sort': 'list (sampling) order even where kernels.ROI_FWD_SORT_MIN asks for the launch order',
    'narrow_mfma_bwd': 'narrow (<= 8 output) heads backward as padded MFMA GEMMs',
}


class _Switches:
    __slots__ = tuple(SWITCHES)

    def __init__(self):
        for k in SWITCHES:
            setattr(self, k, bool(os.environ.get('LOFT_' + k.upper())))

    def active(self):
        return sorted(k for k in SWITCHES if getattr(self, k))

    @contextlib.contextmanager
    def override(self, **kw):
        prev = {k: getattr(self, k) for k in kw}      # (AttributeError for an unknown name: __slots__)
        try:
            for k, v in kw.items():
                setattr(self, k, bool(v))
            yield self
        finally:
            for k, v in prev.items():
                setattr(self, k, v)


DBG = _Switches()
