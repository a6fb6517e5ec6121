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

Feeding the step (round 4): ``batches(..., prefetch=N)`` decodes on a pool of host threads (PIL releases the GIL while it
inflates a tile), writes the decoded tiles straight into a ring of PINNED staging buffers and uploads + normalises + rasterises
on a side HIP stream, N batches ahead of the training step -- the reference's ``workers_per_gpu`` DataLoader processes
(mmdet/datasets/builder.py:58-136) as threads of the one process that owns the GPU.  Every random decision (flip draw,
``_rand_another`` replacement) is taken by the single producer thread in sample order, so the stream of batches is identical to
the synchronous loader's for the same seed.

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
        self.data_infos, self.anns, self.cat_ids, self._ann_cache = [], [], None, {}
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
        if getattr(self, 'cat_ids', None) is not None and list(self.cat_ids) != cat_ids:
            # (the reference builds one dataset PER annotation file and concatenates them, each with its own category map;
            #  one shared map is only right when the files agree -- BONAI's do: a single 'building' category)
            raise ValueError(f'{ann_file}: category ids {cat_ids} differ from the previous files\' {list(self.cat_ids)}')
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
        """bonai.py:105-256 on image ``idx``.  Parsed once and kept (the reference parses on every access; the result is a pure
        function of the annotation file): the loader reads it twice per sample -- resolve() and prepare_train_img()."""
        a = self._ann_cache.get(idx)
        if a is None:
            a = self._ann_cache[idx] = parse_bonai_annotations(self.data_infos[idx], self.anns[idx], cat_ids=tuple(self.cat_ids),
                                                               cat2label=self.cat2label, **self.kw)
        return a

    # ------------------------------------------------------------------ samples
    def _read_image(self, info):
        from PIL import Image
        path = os.path.join(info['_prefix'], info['filename'])
        rgb = np.asarray(Image.open(path).convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])                    # BGR, as mmcv.imread / cv2 deliver it

    def prepare_train_img(self, idx, flip_draw=None, img_out=None, decode=True):
        """One training sample.  flip_draw: the uniform draw that decides the flip (None: drawn here from self.rng).
        decode=False (with img_out): everything but the pixels -- a decoder process fills img_out (decode_tile_into).
        img_out: optional uint8 [H, W, 3] array (a pinned staging slot of the prefetcher).  The decoder's RGB output then goes
        there with ONE copy and nothing else touches the pixels on the host: the sample says ``img_rgb`` (no BGR round trip:
        to_device_batch skips Normalize's to_rgb reversal) and ``img_flip`` (RandomFlip's mirror of the image happens on the
        device after the upload) -- same device values as the host path, a third of the host work and of the time spent holding
        the interpreter lock next to the training loop."""
        info = self.data_infos[idx]
        ann = self.get_ann_info(idx)
        if ann['bboxes'].shape[0] == 0:                                 # custom.py:188-191: no gt after parsing -> another sample
            return None
        if img_out is None:
            img = self._read_image(info)
        elif not decode:
            img = img_out
        else:
            from PIL import Image
            im = Image.open(os.path.join(info['_prefix'], info['filename']))
            if im.mode != 'RGB':
                im = im.convert('RGB')
            if (im.height, im.width) == img_out.shape[:2]:
                np.copyto(img_out, np.asarray(im))
                img = img_out
            else:
                img = np.asarray(im)
        h, w = img.shape[:2]
        if (h, w) != self.img_scale[::-1] or h % 32 or w % 32:
            raise NotImplementedError(f"{info['filename']}: {w}x{h} tile; the device path takes the dataset's {self.img_scale} "
                                      'tiles as they are (Resize / Pad of bonai_instance.py:11,14 are identities there)')
        sample = dict(img=img, filename=info['filename'], gt_bboxes=ann['bboxes'], gt_labels=ann['labels'],
                      gt_offsets=ann['offsets'])
        if img_out is not None:
            sample['img_rgb'] = True
        if self.host_rasteriser is None:
            sample['gt_polygons'] = ann['masks']
            if img_out is not None:                 # the prefetching loader: vertex arrays for loft_poly2mask, packed once per image
                pk = ann.get('_packed')
                if pk is None:
                    from .kernels import pack_polygons
                    pk = ann['_packed'] = pack_polygons(ann['masks'])
                sample['gt_polygons_packed'] = pk
        else:
            sample['gt_masks'] = np.stack([self.host_rasteriser(m, h, w) for m in ann['masks']])
        if self.flip_ratio and (self.rng.rand() if flip_draw is None else flip_draw) < self.flip_ratio:
            sample = flip_sample(sample, self.flip_direction, defer_image=img_out is not None)
        return sample

    def resolve(self, idx):
        """The random decisions of one training sample, in the reference's order: while the sample has no ground truth after
        parsing, another one of its aspect-ratio group (custom.py:170-191); then the flip draw (transforms.py:379-391).
        -> (index actually used, uniform draw for the flip)."""
        while self.get_ann_info(idx)['bboxes'].shape[0] == 0:
            pool = np.where(self.flag == self.flag[idx])[0]
            idx = int(self.rng.choice(pool))
        return idx, (float(self.rng.rand()) if self.flip_ratio else 1.0)

    def __getitem__(self, idx):
        if self.test_mode:
            info = self.data_infos[idx]
            return dict(img=self._read_image(info), filename=info['filename'])
        idx, draw = self.resolve(idx)
        return self.prepare_train_img(idx, flip_draw=draw)

    # ------------------------------------------------------------------ batches
    def epoch_indices(self, epoch, samples_per_gpu, rank=0, world=1, shuffle=True, seed=0):
        """This rank's sample indices for one epoch, a multiple of samples_per_gpu long; same permutation on every rank, padded by
        wrapping so that all ranks run the same number of steps (they meet at the gradient all-reduce every step)."""
        n = len(self)
        order = np.random.RandomState(seed + epoch).permutation(n) if shuffle else np.arange(n)
        per_rank = -(-n // (world * samples_per_gpu)) * samples_per_gpu
        total = per_rank * world
        if total > n:
            # (wrap as often as needed: a small dataset with many ranks / a large batch needs more than one extra pass;
            #  order[:total - n] alone came up short there and left ranks with unequal slices -- a deadlock at the all-reduce)
            order = np.resize(order, total)
        return order[rank * per_rank:(rank + 1) * per_rank].tolist()

    def batches(self, epoch, samples_per_gpu, rank=0, world=1, shuffle=True, seed=0, device='cuda', prefetch=0, workers=8,
                processes=None):
        """Device batches of one epoch for this rank (img, img_metas, gt_bboxes, gt_labels, gt_masks, gt_offsets).
        prefetch = 0: decode + upload synchronously in the caller's thread and stream; prefetch = N > 0: N batches ahead on
        ``workers`` decoder threads, pinned staging and a side stream (see the module docstring) -- same batches, same order."""
        idx = self.epoch_indices(epoch, samples_per_gpu, rank, world, shuffle, seed)
        groups = [idx[i:i + samples_per_gpu] for i in range(0, len(idx), samples_per_gpu)]
        if prefetch <= 0 or self.test_mode:
            for g in groups:
                yield to_device_batch([self[j] for j in g], device=device)
            return
        yield from _Prefetcher(self, groups, device, depth=prefetch, workers=workers, processes=processes)

    # ---- decoder processes and their staging block live as long as the dataset (forking a process that has a HIP context mapped
    # costs ~0.1 s per worker: paid once, not per epoch)
    def _decoder_pool(self, workers):
        pool = getattr(self, '_pool', None)
        if pool is None or self._pool_workers != workers:
            if pool is not None:
                pool.shutdown(wait=True, cancel_futures=True)
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            self._pool = ProcessPoolExecutor(workers, mp_context=mp.get_context('fork'))
            self._pool_workers = workers
            list(self._pool.map(_worker_ready, range(workers)))          # fork them all now
        return self._pool

    def _staging_block(self, nslots, bs, h, w, cuda):
        import torch
        st = getattr(self, '_staging', None)
        if st is not None and st['shape'] == (nslots, bs, h, w) and st['cuda'] == cuda:
            return st
        self._release_staging()
        from multiprocessing import shared_memory
        nbytes = nslots * bs * h * w * 3
        # The block lives in /dev/shm (tmpfs): a container's default 64 MiB does not hold 8 x 1024^2 tiles four deep, and writing
        # past the mount's capacity is a SIGBUS in a decoder process, not an exception.  Check the free space first and create
        # + touch the block here, in the parent, under a guard: None -> the caller takes the thread / pinned-tensor path.
        try:
            vfs = os.statvfs('/dev/shm')
            if vfs.f_bavail * vfs.f_frsize < nbytes + (8 << 20):
                raise OSError(f'/dev/shm has {vfs.f_bavail * vfs.f_frsize >> 20} MiB free, the staging ring needs {nbytes >> 20}')
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
        except (OSError, ValueError) as e:
            import warnings
            warnings.warn(f'decoder processes disabled ({e}); falling back to decoder threads with pinned staging tensors')
            return None
        whole = torch.from_numpy(np.ndarray((nslots, bs, h, w, 3), dtype=np.uint8, buffer=shm.buf))
        registered = None
        if cuda:
            try:                                            # pinned in place: the H2D copy of a slot is one asynchronous DMA
                if int(torch.cuda.cudart().cudaHostRegister(whole.data_ptr(), whole.numel(), 0)) == 0:
                    registered = whole.data_ptr()
            except Exception:                               # noqa -- unpinned staging still works (the copy is then synchronous)
                registered = None
        self._staging = dict(shm=shm, whole=whole, registered=registered, shape=(nslots, bs, h, w), cuda=cuda)
        return self._staging

    def _release_staging(self):
        st = getattr(self, '_staging', None)
        if st is None:
            return
        import torch
        if st['cuda']:
            torch.cuda.synchronize()
            if st['registered'] is not None:
                try:
                    torch.cuda.cudart().cudaHostUnregister(st['registered'])
                except Exception:                           # noqa
                    pass
        st['whole'] = None
        try:
            st['shm'].close()
        except BufferError:
            pass
        try:
            st['shm'].unlink()
        except FileNotFoundError:
            pass
        self._staging = None

    def close(self):
        """Stop the decoder processes and free the staging block (also done when the dataset is collected)."""
        pool = getattr(self, '_pool', None)
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
            self._pool = None
        self._release_staging()

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa
            pass

    def test_batches(self, device='cuda'):
        """Test mode, samples_per_gpu = 1 (mmdet/apis/test.py:26 with the test pipeline of bonai_instance.py:18-31: one scale,
        no flip): yields (idx, dict(img=[tensor 1x3xHxW], img_metas=[[meta]])) in dataset order."""
        for i in range(len(self)):
            info = self.data_infos[i]
            b = to_device_batch([dict(img=self._read_image(info), filename=info['filename'], gt_bboxes=np.zeros((0, 4), np.float32),
                                      gt_labels=np.zeros((0,), np.int64), gt_masks=np.zeros((0, 1, 1), np.uint8),
                                      gt_offsets=np.zeros((0, 2), np.float32))], device=device)
            yield i, dict(img=[b['img']], img_metas=[b['img_metas']])


_SHM_CACHE = {}


def _worker_ready(i):
    import time
    time.sleep(0.05)          # (keeps the first `workers` tasks on distinct processes: every worker is forked at pool creation)
    return i


def decode_tile_into(path, shm_name, offset, h, w):
    """Decoder-process task: decode ``path`` to RGB and write it at byte ``offset`` of the shared-memory block ``shm_name`` as
    uint8 [h, w, 3].  -> None, or a message when the tile has another size.  Touches no torch / HIP state."""
    from multiprocessing import shared_memory
    from PIL import Image
    shm = _SHM_CACHE.get(shm_name)
    if shm is None:
        if len(_SHM_CACHE) > 4:
            for v in _SHM_CACHE.values():
                v.close()
            _SHM_CACHE.clear()
        # (Attaching registers the name with the resource tracker a second time.  The forked workers SHARE the parent's tracker
        #  and its registry is a set per resource type, so the duplicate is harmless -- unregistering it here removed the parent's
        #  own entry: KeyError tracebacks at the parent's unlink() and a leaked block if the parent crashed.  ADVICE round 4.)
        shm = _SHM_CACHE[shm_name] = shared_memory.SharedMemory(name=shm_name)
    im = Image.open(path)
    if im.mode != 'RGB':
        im = im.convert('RGB')
    if (im.height, im.width) != (h, w):
        return f'{path}: {im.width}x{im.height} tile, expected {w}x{h}'
    dst = np.ndarray((h, w, 3), dtype=np.uint8, buffer=shm.buf, offset=offset)
    np.copyto(dst, np.asarray(im))
    return None


class _Prefetcher:
    """Iterator over device batches produced ``depth`` batches ahead of the consumer (BonaiDataset.batches).

    Decoding runs in worker PROCESSES when the target is a GPU (``processes=None`` -> auto): the training loop holds the
    interpreter lock for most of a step (it issues ~550 launches from Python), so decoder THREADS of the same process -- PIL drops
    the lock only inside the inflate loop -- were measured to stretch a 26 ms step to 37 ms while sustaining 390 img/s on their
    own.  The workers are forked like torch DataLoader's, never touch HIP, and write straight into one shared-memory block that
    the parent has registered as pinned host memory (hipHostRegister), so the upload is still one asynchronous copy per batch."""

    def __init__(self, ds, groups, device, depth=2, workers=8, processes=None):
        import queue
        import threading
        import torch
        self.ds, self.groups, self.device, self.depth = ds, groups, torch.device(device), max(1, int(depth))
        self.cuda = self.device.type == 'cuda'
        if self.cuda and self.device.index is None:          # the consumer's current device, fixed now (the producer is another thread)
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.q = queue.Queue(maxsize=self.depth)
        self.stop = threading.Event()
        self.workers = max(1, int(workers))
        self.processes = self.cuda if processes is None else bool(processes)
        bs = max(len(g) for g in groups) if groups else 0
        w, h = ds.img_scale
        # ring of staging buffers: depth in the queue + one being filled + one the consumer's upload may still read
        self.slots, self.shm = [], None
        nslots = self.depth + 2
        st = ds._staging_block(nslots, bs, h, w, self.cuda) if self.processes and bs else None
        if st is None:
            self.processes = False                    # (no shared-memory ring: /dev/shm too small -- decoder threads instead)
        if st is not None:
            self.shm, self.slot_bytes = st['shm'], bs * h * w * 3
            self.slots = [st['whole'][k] for k in range(nslots)]
            self.pool = ds._decoder_pool(self.workers)
        else:
            for _ in range(nslots):
                t = torch.empty((bs, h, w, 3), dtype=torch.uint8)
                self.slots.append(t.pin_memory() if self.cuda else t)
        self.slot_free = [None] * len(self.slots)         # event after which a slot's upload has completed
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.thread = threading.Thread(target=self._produce, name='bonai-prefetch', daemon=True)
        self.thread.start()

    def _produce(self):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            import contextlib
            pool_cm = contextlib.nullcontext(self.pool) if self.processes else ThreadPoolExecutor(self.workers, thread_name_prefix='bonai-decode')
            with pool_cm as pool:
                for bi, g in enumerate(self.groups):
                    if self.stop.is_set():
                        return
                    k = bi % len(self.slots)
                    if self.slot_free[k] is not None:
                        self.slot_free[k].synchronize()            # the upload that last read this slot has finished
                    buf = self.slots[k].numpy()
                    # every random decision in THIS thread, in the synchronous loader's order -- per sample: replacements of a
                    # sample without ground truth (custom.py:188-191), then its flip draw -- so both loaders emit the same stream
                    futs, samples = [], []
                    h, w = buf.shape[1:3]
                    for i, j in enumerate(g):
                        j, draw = self.ds.resolve(j)
                        if self.processes:
                            info = self.ds.data_infos[j]
                            futs.append(pool.submit(decode_tile_into, os.path.join(info['_prefix'], info['filename']), self.shm.name,
                                                    k * self.slot_bytes + i * h * w * 3, h, w))
                            samples.append(self.ds.prepare_train_img(j, draw, buf[i], decode=False))
                        else:
                            futs.append(pool.submit(self.ds.prepare_train_img, j, draw, buf[i]))
                    if self.processes:
                        for f in futs:
                            err = f.result()
                            if err is not None:
                                raise NotImplementedError(err + ' (the device path takes fixed-size tiles)')
                    else:
                        samples = [f.result() for f in futs]
                    if self.cuda:
                        with torch.cuda.stream(self.side):
                            batch = to_device_batch(samples, device=self.device, staged=self.slots[k][:len(g)])
                            ev = torch.cuda.Event()
                            ev.record(self.side)
                        self.slot_free[k] = ev
                    else:
                        batch, ev = to_device_batch(samples, device=self.device), None
                    while not self.stop.is_set():
                        try:
                            self.q.put((batch, ev), timeout=0.1)
                            break
                        except Exception:
                            continue
            self.q.put((None, None))
        except BaseException as e:      # noqa -- surfaced in the consumer
            self.q.put((e, None))

    def __iter__(self):
        import torch
        try:
            while True:
                batch, ev = self.q.get()
                if batch is None:
                    return
                if isinstance(batch, BaseException):
                    raise batch
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for v in batch.values():                       # allocated on the side stream, consumed on this one
                        for t in (v if isinstance(v, (list, tuple)) else [v]):
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(cur)
                yield batch
        finally:
            self.stop.set()
            while self.thread.is_alive():                          # unblock a producer waiting on a full queue
                try:
                    self.q.get_nowait()
                except Exception:
                    pass
                self.thread.join(0.05)
            self._release()

    def _release(self):
        import torch
        if self.cuda:
            torch.cuda.synchronize(self.device)                    # no upload may still be reading the staging ring
        self.slots = []
