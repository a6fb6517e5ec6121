# BONAI instance data contract (reference: configs/_base_/datasets/bonai_instance.py).  The CPU data pipeline is
# out of scope for the MI355X hot path; only the batch-dict keys are kept:
#   img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_offsets
dataset_type = 'BONAI'
data_root = 'data/BONAI/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
batch_keys = ['img', 'img_metas', 'gt_bboxes', 'gt_labels', 'gt_masks', 'gt_offsets']
data = dict(samples_per_gpu=8, workers_per_gpu=2)
evaluation = dict(interval=1, metric=['bbox', 'segm'])
