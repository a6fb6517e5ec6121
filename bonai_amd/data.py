"""BONAI sample contract on the host side of the path: annotation schema -> training sample -> device batch.

What the hot path consumes is the batch dict ``img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_offsets``
(mmdet/models/detectors/two_stage.py:105-167).  This module mirrors the pieces of the reference's data pipeline that
define the *values* in that dict -- the BONAI annotation parser (mmdet/datasets/bonai.py:105-256), the flip rules for
boxes and offsets (mmdet/datasets/pipelines/transforms.py:379-404, 458-466), Normalize and DefaultFormatBundle / Collect
(pipelines/formating.py) -- and ends in ``to_device_batch``, which uploads once: images normalised on the GPU, instance
masks as uint8 device tensors (the device-side mask_target kernel crops them; no per-step CPU round trip as in
mmdet/core/mask/structures.py:261-291).  Polygon rasterisation (pycocotools in the reference) runs on the device
(kernels.poly2mask); image decoding (cv2) stays outside: it is not in this image and not on the path.
"""
import math

import numpy as np
import torch


def parse_bonai_annotations(img_info, ann_info, cat_ids=(1,), cat2label=None, bbox_type='roof', mask_type='roof',
                            offset_coordinate='rectangle', resolution=0.6, ignore_buildings=True):
    """BONAI._parse_ann_info (bonai.py:105-256): list of COCO-style BONAI annotation dicts -> ann dict of arrays.
    Keys and dtypes follow the reference, including its empty-image conventions (angle 0.0001, heights (0, 2))."""
    cat2label = cat2label or {c: i for i, c in enumerate(cat_ids)}
    key = {'roof': 'bbox', 'building': 'building_bbox', 'footprint': 'footprint_bbox'}
    if bbox_type not in key:
        raise TypeError(f"don't support bbox_type={bbox_type}")
    if mask_type not in ('roof', 'footprint'):
        raise TypeError(f"don't support mask_type={mask_type}")
    if offset_coordinate not in ('rectangle', 'polar'):
        raise RuntimeError(f'do not support this coordinate: {offset_coordinate}')
    bboxes, labels, ignore, masks, roof_masks, fp_masks = [], [], [], [], [], []
    offsets, heights, angles, roof_bboxes, fp_bboxes = [], [], [], [], []
    only_fp = 0
    for ann in ann_info:
        if ann.get('ignore', False):
            continue
        x1, y1, w, h = ann[key[bbox_type]]
        inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
        inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
        if inter_w * inter_h == 0 or ann['area'] <= 0 or w < 1 or h < 1 or ann['category_id'] not in cat_ids:
            continue
        bbox = [x1, y1, x1 + w, y1 + h]
        if ann.get('iscrowd', False) and ignore_buildings:
            ignore.append(bbox)
            continue
        if 'roof_bbox' in ann:
            rx, ry, rw, rh = ann['roof_bbox']
            roof_bboxes.append([rx, ry, rx + rw, ry + rh])
        if 'footprint_bbox' in ann:
            fx, fy, fw, fh = ann['footprint_bbox']
            fp_bboxes.append([fx, fy, fx + fw, fy + fh])
        if 'only_footprint' in ann:
            only_fp = 1 if ann['only_footprint'] == 1 else 0
        bboxes.append(bbox)
        labels.append(cat2label[ann['category_id']])
        if only_fp == 0 and mask_type == 'roof':
            masks.append(ann['segmentation'])
        else:
            masks.append([ann['footprint_mask']])
        roof_masks.append(ann['segmentation'])
        fp_masks.append([ann['footprint_mask']])
        if 'offset' in ann:
            if offset_coordinate == 'rectangle':
                offsets.append(ann['offset'])
            else:
                ox, oy = ann['offset']
                offsets.append([math.sqrt(ox ** 2 + oy ** 2), math.atan2(oy, ox)])
        else:
            offsets.append([0, 0])
        heights.append(ann.get('building_height', 0.0))
        if 'offset' in ann and 'building_height' in ann:
            ox, oy = ann['offset']
            angles.append(math.atan2(math.sqrt(ox ** 2 + oy ** 2) * resolution, ann['building_height']))
    if bboxes:
        out = dict(bboxes=np.array(bboxes, dtype=np.float32), labels=np.array(labels, dtype=np.int64),
                   offsets=np.array(offsets, dtype=np.float32), building_heights=np.array(heights, dtype=np.float32),
                   angle=float(np.array(angles, dtype=np.float32).mean()) if angles else float('nan'),
                   roof_bboxes=np.array(roof_bboxes, dtype=np.float32), footprint_bboxes=np.array(fp_bboxes, dtype=np.float32),
                   only_footprint_flag=float(only_fp))
    else:
        out = dict(bboxes=np.zeros((0, 4), np.float32), labels=np.array([], dtype=np.int64), offsets=np.zeros((0, 2), np.float32),
                   building_heights=np.zeros((0, 2), np.float32), angle=0.0001, roof_bboxes=np.zeros((0, 4), np.float32),
                   footprint_bboxes=np.zeros((0, 4), np.float32), only_footprint_flag=0)
    out['bboxes_ignore'] = np.array(ignore, dtype=np.float32) if ignore else np.zeros((0, 4), np.float32)
    out.update(masks=masks, roof_masks=roof_masks, footprint_masks=fp_masks)
    fn = img_info['filename']
    out.update(seg_map=fn.replace('jpg', 'png'), edge_map=fn.replace('jpg', 'png'), side_face_map=fn.replace('jpg', 'png'),
               offset_field=fn.replace('png', 'npy'))
    return out


def flip_bboxes(bboxes, img_shape, direction):
    """RandomFlip.bbox_flip (transforms.py:379-404)."""
    out = bboxes.copy()
    if direction == 'horizontal':
        w = img_shape[1]
        out[..., 0::4] = w - bboxes[..., 2::4]
        out[..., 2::4] = w - bboxes[..., 0::4]
    elif direction == 'vertical':
        h = img_shape[0]
        out[..., 1::4] = h - bboxes[..., 3::4]
        out[..., 3::4] = h - bboxes[..., 1::4]
    else:
        raise ValueError(f"Invalid flipping direction '{direction}'")
    return out


def flip_offsets(offsets, direction):
    """RandomFlip.offset_flip (transforms.py:458-466): horizontal negates x, vertical negates y."""
    off = np.asarray(offsets, dtype=np.float32).reshape(-1, 2).copy()
    if direction == 'horizontal':
        off[:, 0] = -off[:, 0]
    elif direction == 'vertical':
        off[:, 1] = -off[:, 1]
    else:
        raise ValueError(f"Invalid flipping direction '{direction}'")
    return off


def flip_sample(sample, direction='horizontal', defer_image=False):
    """One training sample (img HxWx3, gt_bboxes, gt_masks [K,H,W] u8 -- or gt_polygons --, gt_offsets) flipped as
    RandomFlip.__call__ does (transforms.py:406-456).

    A sample that carries ``gt_polygons`` instead of bitmaps keeps its polygons as they are and records the flip in
    ``mask_flips``: the reference's pipeline rasterises first (LoadAnnotations(poly2mask=True), loading.py:301-326 ->
    BitmapMasks) and RandomFlip then mirrors the BITMAP (structures.py:218-229), which is not the same pixels as rasterising
    mirrored vertices (pycocotools' edge walk is not flip-symmetric); ``to_device_batch`` rasterises on the device and
    mirrors the bitmap there, in the recorded order -- bit-identical to the host bitmap path."""
    if direction not in ('horizontal', 'vertical'):
        raise ValueError(f"Invalid flipping direction '{direction}'")
    h, w = sample['img'].shape[:2]
    ax = 1 if direction == 'horizontal' else 0
    out = dict(sample)
    if defer_image:          # the prefetching loader: the image is mirrored on the device after the upload (to_device_batch)
        out['img_flip'] = tuple(sample.get('img_flip', ())) + (direction,)
    else:
        out['img'] = np.flip(sample['img'], axis=ax).copy()
    out['gt_bboxes'] = flip_bboxes(sample['gt_bboxes'], (h, w), direction)
    if sample.get('gt_masks') is not None:
        out['gt_masks'] = np.flip(sample['gt_masks'], axis=ax + 1).copy()
    elif 'gt_polygons' in sample:
        out['mask_flips'] = tuple(sample.get('mask_flips', ())) + (direction,)
    else:
        raise KeyError("flip_sample: the sample carries neither 'gt_masks' nor 'gt_polygons'")
    out['gt_offsets'] = flip_offsets(sample['gt_offsets'], direction)
    out['flip'], out['flip_direction'] = True, direction
    return out


def _masks_of(s, dev):
    if 'gt_polygons' in s and s.get('gt_masks') is None:
        from . import kernels as K
        h, w = s['img'].shape[:2]
        m = K.poly2mask(s.get('gt_polygons_packed') or s['gt_polygons'], h, w, device=dev)
        for d in s.get('mask_flips', ()):                   # flip_sample on a polygon sample: the bitmap is mirrored, as the
            m = m.flip(2 if d == 'horizontal' else 1)       # reference's RandomFlip does after LoadAnnotations rasterised it
        return m.contiguous()
    return torch.from_numpy(np.ascontiguousarray(s['gt_masks'], dtype=np.uint8)).to(dev)


def to_device_batch(samples, device='cuda', mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True, staged=None):
    """Collate + DefaultFormatBundle + Normalize, on the device: list of samples (img uint8/float HxWx3 BGR, gt_* numpy) ->
    the batch dict of forward_train.  Images are stacked (same size: BONAI tiles are 1024x1024), normalised on the GPU;
    ``staged``: the images as one uint8 [n,H,W,3] host tensor (BonaiDataset's pinned staging ring; sample['img'] are views of it).
    masks go up once as uint8 [K,H,W] tensors -- or, when a sample carries ``gt_polygons`` (per instance a list of flat polygons: the
    annotation's ``masks`` entry, bonai.py:186-199) instead of ``gt_masks``, only the vertices go up and the bitmaps are rasterised on the
    device (kernels.poly2mask = LoadAnnotations._poly2mask, loading.py:301-326): no K x 1024^2 host bitmaps, no upload."""
    dev = torch.device(device)
    if staged is not None:       # uint8 [n, H, W, 3] holding the samples' images already (a pinned staging slot): one async upload
        imgs = staged.to(dev, non_blocking=True)
    else:
        imgs = torch.stack([torch.from_numpy(np.ascontiguousarray(s['img'])) for s in samples]).to(dev)
    x = imgs.float()
    rgb = [bool(s.get('img_rgb', False)) for s in samples]         # decoded straight to RGB: Normalize's reversal already done
    if to_rgb and not all(rgb):
        if any(rgb):
            keep = torch.tensor(rgb, device=dev).view(-1, 1, 1, 1)
            x = torch.where(keep, x, x.flip(-1))
        else:
            x = x.flip(-1)
    elif not to_rgb and any(rgb):
        raise ValueError('samples decoded to RGB need to_rgb=True (the configured Normalize of bonai_instance.py:3-4)')
    for i, s in enumerate(samples):                                  # RandomFlip's image mirror, deferred by the loader
        for d in s.get('img_flip', ()):
            x[i] = x[i].flip(1 if d == 'horizontal' else 0)
    x = (x - torch.tensor(mean, device=dev)) / torch.tensor(std, device=dev)
    img = x.permute(0, 3, 1, 2).contiguous()
    metas = []
    for s in samples:
        h, w = s['img'].shape[:2]
        metas.append(dict(filename=s.get('filename'), ori_shape=(h, w, 3), img_shape=(h, w, 3), pad_shape=(h, w, 3),
                          scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), flip=bool(s.get('flip', False)),
                          flip_direction=s.get('flip_direction'),
                          img_norm_cfg=dict(mean=np.array(mean, np.float32), std=np.array(std, np.float32), to_rgb=to_rgb)))
    if dev.type == 'cuda' and staged is not None:
        # the loader's path: every small array of the batch (boxes, labels, offsets, polygon vertices and their offset tables)
        # in ONE pinned buffer and ONE asynchronous copy -- 48 pageable synchronous copies of a few hundred bytes each were
        # 5 of the 6 ms this function held the interpreter lock next to the training loop
        return dict(img=img, img_metas=metas, **_small_arrays_one_copy(samples, dev))
    return dict(img=img, img_metas=metas,
                gt_bboxes=[torch.from_numpy(np.asarray(s['gt_bboxes'], np.float32)).to(dev) for s in samples],
                gt_labels=[torch.from_numpy(np.asarray(s['gt_labels'], np.int64)).to(dev) for s in samples],
                gt_masks=[_masks_of(s, dev) for s in samples],
                gt_offsets=[torch.from_numpy(np.asarray(s['gt_offsets'], np.float32).reshape(-1, 2)).to(dev) for s in samples])


def _small_arrays_one_copy(samples, dev):
    from . import kernels as K
    arrs, slots = [], []          # (numpy array) and (sample index, field)
    packs = []
    for i, s in enumerate(samples):
        arrs += [np.ascontiguousarray(s['gt_bboxes'], np.float32), np.ascontiguousarray(s['gt_labels'], np.int64),
                 np.ascontiguousarray(np.asarray(s['gt_offsets'], np.float32).reshape(-1, 2))]
        slots += [(i, 'gt_bboxes'), (i, 'gt_labels'), (i, 'gt_offsets')]
        pk = None
        if 'gt_polygons' in s and s.get('gt_masks') is None:
            pk = s.get('gt_polygons_packed') or K.pack_polygons(s['gt_polygons'])
            arrs += [pk.xy, pk.poff, pk.ioff]
            slots += [(i, '_xy'), (i, '_poff'), (i, '_ioff')]
        packs.append(pk)
    offs, total = [], 0
    for a in arrs:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, o in zip(arrs, offs):
        if a.nbytes:
            hv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    devb = host.to(dev, non_blocking=True)
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int64): torch.int64, np.dtype(np.float64): torch.float64}
    out = dict(gt_bboxes=[None] * len(samples), gt_labels=[None] * len(samples), gt_offsets=[None] * len(samples))
    aux = [dict() for _ in samples]
    for a, o, (i, name) in zip(arrs, offs, slots):
        t = devb[o:o + a.nbytes].view(tdt[a.dtype]).view(a.shape)
        if name[0] == '_':
            aux[i][name] = t
        else:
            out[name][i] = t
    masks = []
    for i, s in enumerate(samples):
        if packs[i] is None:
            masks.append(_masks_of(s, dev))
            continue
        h, w = s['img'].shape[:2]
        m = K.poly2mask_device(aux[i]['_xy'], aux[i]['_poff'], aux[i]['_ioff'], packs[i].n, h, w, packs[i].maxv)
        for d in s.get('mask_flips', ()):
            m = m.flip(2 if d == 'horizontal' else 1)
        masks.append(m.contiguous())
    out['gt_masks'] = masks
    return out
