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
# BASELINE config 5 is the reference's fp16 recipe (configs/fp16/*: Fp16OptimizerHook, static loss scale): bench.py and
# tools/train.py switch to the binary16 build of the kernels (bonai_amd.lib.set_act16(torch.float16) -> libloft_hip_f16.so,
# v_mfma_f32_32x32x16_f16) and bonai_amd.engine.Trainer applies the static scale; masters, losses, RoIAlign, coders stay fp32.
fp16 = dict(loss_scale=512.)
