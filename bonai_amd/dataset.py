"""BONAI annotation files -> training samples -> device batches (missing #6).

The reference reaches the hot path through ``mmdet.datasets.BONAI`` (a ``CocoDataset``: mmdet/datasets/bonai.py:14-104,
coco.py:36-78, custom.py:115-213) and the train pipeline of configs/_base_/datasets/bonai_instance.py:5-17
(LoadImageFromFile, LoadAnnotations(with_bbox, with_mask, with_offset), Resize((1024, 1024), keep_ratio), RandomFlip(0.5,
['horizontal', 'vertical']), Normalize, Pad(32), DefaultFormatBundle, Collect).  This module is the part of that chain that
decides WHICH values enter the batch dict; the values themselves are produced by bonai_amd/data.py (parser, flip rules,
``to_device_batch`` = Normalize + bundle + collate on the device, polygons rasterised by kernels.poly2mask).

What is mirrored, with the reference's semantics:
  * COCO-style index without pycocotools: images, annotations per image, category ids of CLASSES = ('building')
    (coco.py:46-55; bonai.py:16 -- the tuple-less string is the reference's own: every character-free lookup ends at 'building');
  * ``_filter_imgs`` (bonai.py:85-100): training drops images without annotations or with only crowd annotations and images
    smaller than 32 px;
  * ``_rand_another`` (custom.py:170-191): a sample that ends up without ground truth after parsing is replaced by a random one
    of the same aspect-ratio group;
  * RandomFlip: probability ``flip_ratio`` per sample; a direction LIST is resolved once, when the pipeline is built
    (transforms.py:367-377: ``np.random.choice(direction)`` in ``__init__``) -- every flipped sample of a run uses that direction;
  * Resize / Pad: BONAI tiles are 1024 x 1024, the configured scale is (1024, 1024) with keep_ratio and the pad divisor 32, so both
    are identities; another tile size raises (the device path takes fixed-size batches).
Image decoding uses PIL (cv2 / mmcv are not in this image); ``mmcv.imread`` returns BGR, so the RGB decode is reversed to BGR and
``to_device_batch(to_rgb=True)`` converts it back exactly as Normalize does.

Data parallelism: ``epoch_indices`` is DistributedGroupSampler's contract (datasets/builder.py:107-110): every rank gets a disjoint,
equally long slice of a per-epoch permutation seeded identically on all ranks.
"""
import json
import os

import numpy as np

from .data import flip_sample, parse_bonai_annotations, to_device_batch

CLASSES = ('building',)


class BonaiDataset:
    def __init__(self, ann_file, img_prefix='', classes=None, test_mode=False, filter_empty_gt=True, bbox_type='roof',
                 mask_type='roof', offset_coordinate='rectangle', resolution=0.6, ignore_buildings=True, flip_ratio=0.5,
                 flip_direction=('horizontal', 'vertical'), img_scale=(1024, 1024), seed=0, host_rasteriser=None):
        ann_files = [ann_file] if isinstance(ann_file, str) else list(ann_file)
        prefixes = [img_prefix] * len(ann_files) if isinstance(img_prefix, str) else list(img_prefix)
        if len(prefixes) != len(ann_files):
            raise ValueError('ann_file and img_prefix lists must have the same length (bonai_instance.py:33-38)')
        self.classes = tuple(classes) if classes is not None else CLASSES
        self.test_mode, self.filter_empty_gt = test_mode, filter_empty_gt
        self.kw = dict(bbox_type=bbox_type, mask_type=mask_type, offset_coordinate=offset_coordinate, resolution=resolution,
                       ignore_buildings=ignore_buildings)
        self.flip_ratio, self.img_scale = flip_ratio, tuple(img_scale)
        # host_rasteriser(polygons_of_one_instance, h, w) -> uint8 [h, w]: a caller-supplied host rasteriser (the tests pass the
        # oracle's); None (the product): polygons travel to the device and kernels.poly2mask rasterises them there
        self.host_rasteriser = host_rasteriser
        self.rng = np.random.RandomState(seed)
        if isinstance(flip_direction, str):
            self.flip_direction = flip_direction
        else:                                           # the reference resolves a list ONCE (transforms.py:371-372)
            self.flip_direction = str(self.rng.choice(list(flip_direction)))
        if self.flip_direction not in ('horizontal', 'vertical'):
            raise ValueError(f"Invalid flipping direction '{self.flip_direction}'")
        self.data_infos, self.anns = [], []
        for f, prefix in zip(ann_files, prefixes):
            self._load(f, prefix)
        if not test_mode:
            keep = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in keep]
            self.anns = [self.anns[i] for i in keep]
        # custom.py:158-168: group flag by aspect ratio
        self.flag = np.array([1 if d['width'] / d['height'] > 1 else 0 for d in self.data_infos], dtype=np.uint8)

    # ------------------------------------------------------------------ index
    def _load(self, ann_file, prefix):
        with open(ann_file) as fh:
            coco = json.load(fh)
        cat_ids = [c['id'] for c in coco.get('categories', []) if c['name'] in self.classes]
        if not cat_ids:
            raise ValueError(f'{ann_file}: no category named {self.classes}')
        self.cat_ids = cat_ids
        self.cat2label = {c: i for i, c in enumerate(cat_ids)}
        by_img = {}
        for a in coco.get('annotations', []):
            by_img.setdefault(a['image_id'], []).append(a)
        for info in coco['images']:
            info = dict(info)
            info['filename'] = info['file_name']
            info['_prefix'] = prefix
            self.data_infos.append(info)
            self.anns.append(by_img.get(info['id'], []))

    def _filter_imgs(self, min_size=32):
        keep = []
        for i, (info, anns) in enumerate(zip(self.data_infos, self.anns)):
            all_crowd = all(a.get('iscrowd', 0) for a in anns)        # (all([]) is True: an image without annotations drops too)
            if self.filter_empty_gt and (not anns or all_crowd):
                continue
            if min(info['width'], info['height']) >= min_size:
                keep.append(i)
        return keep

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        return parse_bonai_annotations(self.data_infos[idx], self.anns[idx], cat_ids=tuple(self.cat_ids),
                                       cat2label=self.cat2label, **self.kw)

    # ------------------------------------------------------------------ samples
    def _read_image(self, info):
        from PIL import Image
        path = os.path.join(info['_prefix'], info['filename'])
        rgb = np.asarray(Image.open(path).convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])                    # BGR, as mmcv.imread / cv2 deliver it

    def prepare_train_img(self, idx):
        info = self.data_infos[idx]
        ann = self.get_ann_info(idx)
        if ann['bboxes'].shape[0] == 0:                                 # custom.py:188-191: no gt after parsing -> another sample
            return None
        img = self._read_image(info)
        h, w = img.shape[:2]
        if (h, w) != self.img_scale[::-1] or h % 32 or w % 32:
            raise NotImplementedError(f"{info['filename']}: {w}x{h} tile; the device path takes the dataset's {self.img_scale} "
                                      'tiles as they are (Resize / Pad of bonai_instance.py:11,14 are identities there)')
        sample = dict(img=img, filename=info['filename'], gt_bboxes=ann['bboxes'], gt_labels=ann['labels'],
                      gt_offsets=ann['offsets'])
        if self.host_rasteriser is None:
            sample['gt_polygons'] = ann['masks']
        else:
            sample['gt_masks'] = np.stack([self.host_rasteriser(m, h, w) for m in ann['masks']])
        if self.flip_ratio and self.rng.rand() < self.flip_ratio:
            sample = flip_sample(sample, self.flip_direction)
        return sample

    def __getitem__(self, idx):
        if self.test_mode:
            info = self.data_infos[idx]
            return dict(img=self._read_image(info), filename=info['filename'])
        while True:
            s = self.prepare_train_img(idx)
            if s is not None:
                return s
            pool = np.where(self.flag == self.flag[idx])[0]
            idx = int(self.rng.choice(pool))

    # ------------------------------------------------------------------ batches
    def epoch_indices(self, epoch, samples_per_gpu, rank=0, world=1, shuffle=True, seed=0):
        """This rank's sample indices for one epoch, a multiple of samples_per_gpu long; same permutation on every rank, padded by
        wrapping so that all ranks run the same number of steps (they meet at the gradient all-reduce every step)."""
        n = len(self)
        order = np.random.RandomState(seed + epoch).permutation(n) if shuffle else np.arange(n)
        per_rank = -(-n // (world * samples_per_gpu)) * samples_per_gpu
        total = per_rank * world
        order = np.concatenate([order, order[:total - n]]) if total > n else order
        return order[rank * per_rank:(rank + 1) * per_rank].tolist()

    def batches(self, epoch, samples_per_gpu, rank=0, world=1, shuffle=True, seed=0, device='cuda'):
        """Device batches of one epoch for this rank (img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_offsets)."""
        idx = self.epoch_indices(epoch, samples_per_gpu, rank, world, shuffle, seed)
        for i in range(0, len(idx), samples_per_gpu):
            yield to_device_batch([self[j] for j in idx[i:i + samples_per_gpu]], device=device)
