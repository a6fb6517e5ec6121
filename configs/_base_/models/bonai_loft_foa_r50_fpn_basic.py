# LOFT + FOA, ResNet-50-FPN -- model / train / test settings.
# Same keys and values as the reference entry point (configs/_base_/models/bonai_loft_foa_r50_fpn_basic.py);
# every `type` resolves in bonai_amd.loft.builder to an MI355X-native class.
_STRIDES = [4, 8, 16, 32]


def _extractor(size):
    return dict(type='SingleRoIExtractor', out_channels=256, featmap_strides=_STRIDES,
                roi_layer=dict(type='RoIAlign', output_size=size, sampling_ratio=0))


def _assigner(pos, neg, min_pos):
    return dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=min_pos, match_low_quality=True,
                ignore_iof_thr=-1, gpu_assign_thr=512)


def _proposals():
    return dict(nms_across_levels=False, nms_pre=3000, nms_post=3000, max_num=3000, nms_thr=0.7, min_bbox_size=0)


model = dict(
    type='LOFT',
    pretrained='torchvision://resnet50',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
    rpn_head=dict(
        type='RPNHead', in_channels=256, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
    roi_head=dict(
        type='LoftRoIHead',
        bbox_roi_extractor=_extractor(7),
        bbox_head=dict(
            type='Shared2FCBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=1,
            bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
            reg_class_agnostic=False,
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
            loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        mask_roi_extractor=_extractor(14),
        mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256, num_classes=1,
                       loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0)),
        offset_roi_extractor=_extractor(7),
        offset_head=dict(type='OffsetHeadExpandFeature', expand_feature_num=4, share_expand_fc=True,
                         rotations=[0, 90, 180, 270], num_fcs=2, fc_out_channels=1024, num_convs=10,
                         loss_offset=dict(type='SmoothL1Loss', loss_weight=8 * 2.0))))

train_cfg = dict(
    rpn=dict(assigner=_assigner(0.7, 0.3, 0.3),
             sampler=dict(type='RandomSampler', num=512, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
             allowed_border=-1, pos_weight=-1, debug=False),
    rpn_proposal=_proposals(),
    rcnn=dict(assigner=_assigner(0.5, 0.5, 0.5),
              sampler=dict(type='RandomSampler', num=1024, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
              mask_size=28, pos_weight=-1, debug=False))

test_cfg = dict(
    rpn=_proposals(),
    rcnn=dict(score_thr=0.05, nms=dict(type='soft_nms', iou_threshold=0.5), max_per_img=2000, mask_thr_binary=0.5))
