"""Synthetic BONAI-shaped batches (SURVEY.md section 8d "Synthetic inputs"): there is no network for the
dataset, so bench / smoke / parity runs use seeded tiles with the reference's batch-dict keys
(configs/_base_/datasets/bonai_instance.py:16): img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_offsets."""
import numpy as np
import torch


def make_gt(global_img_idx, size=1024, num_gt=80):
    rng = np.random.RandomState(1234 + global_img_idx)
    lo, hi = np.log(16.0 * size / 1024), np.log(160.0 * size / 1024)
    w = np.exp(rng.uniform(lo, hi, num_gt))
    h = np.exp(rng.uniform(lo, hi, num_gt))
    x1 = rng.uniform(0, size - w)
    y1 = rng.uniform(0, size - h)
    boxes = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    masks = np.zeros((num_gt, size, size), np.uint8)
    for i in range(num_gt):
        ix, iy = 0.1 * w[i], 0.1 * h[i]
        xa, xb = int(round(x1[i] + ix)), int(round(x1[i] + w[i] - ix))
        ya, yb = int(round(y1[i] + iy)), int(round(y1[i] + h[i] - iy))
        masks[i, ya:max(yb, ya + 1), xa:max(xb, xa + 1)] = 1
    theta_img = rng.uniform(-np.pi, np.pi)
    theta = theta_img + rng.normal(0, 0.05, num_gt)
    length = rng.uniform(0, 40.0 * size / 1024, num_gt)
    offsets = np.stack([length * np.cos(theta), length * np.sin(theta)], 1).astype(np.float32)
    return boxes, masks, offsets


def make_batch(batch_size, size=1024, num_gt=80, rank=0, step=0, device='cpu'):
    g = torch.Generator().manual_seed(20260928 + rank + 1000 * step)
    img = torch.randn(batch_size, 3, size, size, generator=g, dtype=torch.float32)
    gt_bboxes, gt_labels, gt_masks, gt_offsets = [], [], [], []
    for i in range(batch_size):
        b, m, o = make_gt((rank * 100003 + step) * batch_size + i, size, num_gt)
        gt_bboxes.append(torch.from_numpy(b).to(device))
        gt_labels.append(torch.zeros(num_gt, dtype=torch.long, device=device))
        gt_masks.append(torch.from_numpy(m).to(device))
        gt_offsets.append(torch.from_numpy(o).to(device))
    metas = [dict(img_shape=(size, size, 3), pad_shape=(size, size, 3), ori_shape=(size, size, 3),
                  scale_factor=np.ones(4, np.float32), flip=False) for _ in range(batch_size)]
    return dict(img=img.to(device), img_metas=metas, gt_bboxes=gt_bboxes, gt_labels=gt_labels, gt_masks=gt_masks,
                gt_offsets=gt_offsets)


def synth_bonai_anns(seed=0, n=12, size=1024):
    """Synthetic BONAI annotation dicts exercising every branch of the parser (ignore, crowd, zero-area, outside, missing
    optional keys, only_footprint)."""
    rng = np.random.RandomState(seed)
    anns = []
    for i in range(n):
        w, h = rng.uniform(8, 200, 2)
        x, y = rng.uniform(-20, size - 30, 2)
        ox, oy = rng.uniform(-40, 40, 2)
        a = dict(bbox=[float(x), float(y), float(w), float(h)],
                 building_bbox=[float(x - 5), float(y - 5), float(w + 10 + abs(ox)), float(h + 10 + abs(oy))],
                 footprint_bbox=[float(x + ox), float(y + oy), float(w), float(h)],
                 roof_bbox=[float(x), float(y), float(w), float(h)],
                 segmentation=[[float(x), float(y), float(x + w), float(y), float(x + w), float(y + h), float(x), float(y + h)]],
                 footprint_mask=[float(x + ox), float(y + oy), float(x + ox + w), float(y + oy), float(x + ox + w), float(y + oy + h),
                                 float(x + ox), float(y + oy + h)],
                 area=float(w * h), category_id=1, iscrowd=0, offset=[float(ox), float(oy)], building_height=float(rng.uniform(3, 90)))
        anns.append(a)
    anns[1]['ignore'] = True
    anns[2]['iscrowd'] = 1
    anns[3]['area'] = 0.0
    anns[4]['bbox'] = [2000.0, 2000.0, 50.0, 50.0]
    anns[4]['building_bbox'] = [2000.0, 2000.0, 50.0, 50.0]
    anns[4]['footprint_bbox'] = [2000.0, 2000.0, 50.0, 50.0]
    anns[5]['category_id'] = 7
    del anns[6]['offset']
    del anns[7]['building_height']
    anns[8]['only_footprint'] = 1
    anns[9]['only_footprint'] = 0
    anns[10]['bbox'][2] = 0.5
    return anns
