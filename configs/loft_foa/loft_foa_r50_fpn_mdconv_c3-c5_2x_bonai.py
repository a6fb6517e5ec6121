# BASELINE config 4: LOFT R50-FPN with modulated deformable convolutions (DCNv2) in backbone stages c3-c5
# (the reference's configs/dcn/*_mdconv_c3-c5_* backbone override) and in the FPN neck (FPN conv_cfg, fpn.py:116-132).
_base_ = './loft_foa_r50_fpn_2x_bonai.py'
model = dict(
    backbone=dict(
        dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False),
        stage_with_dcn=(False, True, True, True)),
    neck=dict(conv_cfg=dict(type='DCNv2')))
