# BASELINE config 5: LOFT + FOA on an HRNetV2p-W32 backbone with the HRFPN neck (the reference's
# configs/hrnet/mask_rcnn_hrnetv2p_w32_1x_coco.py backbone/neck override applied to the LOFT model).
_base_ = './loft_foa_r50_fpn_2x_bonai.py'
model = dict(
    pretrained=None,
    backbone=dict(
        _delete_=True,
        type='HRNet',
        extra=dict(
            stage1=dict(num_modules=1, num_branches=1, block='BOTTLENECK', num_blocks=(4, ), num_channels=(64, )),
            stage2=dict(num_modules=1, num_branches=2, block='BASIC', num_blocks=(4, 4), num_channels=(32, 64)),
            stage3=dict(num_modules=4, num_branches=3, block='BASIC', num_blocks=(4, 4, 4), num_channels=(32, 64, 128)),
            stage4=dict(num_modules=3, num_branches=4, block='BASIC', num_blocks=(4, 4, 4, 4),
                        num_channels=(32, 64, 128, 256)))),
    neck=dict(_delete_=True, type='HRFPN', in_channels=[32, 64, 128, 256], out_channels=256))
# BASELINE config 5 is the reference's fp16 recipe (configs/fp16/*: Fp16OptimizerHook, static loss scale).  Activations run
# in bf16 here (dtype >= the reference's fp16); the static loss scale is honoured by bonai_amd.engine.Trainer.
fp16 = dict(loss_scale=512.)
