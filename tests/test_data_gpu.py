"""GPU: the device side of the BONAI data contract.  Polygon -> bitmap rasterisation (kernels.poly2mask replacing
LoadAnnotations._poly2mask, mmdet/datasets/pipelines/loading.py:301-326) bit-exact against the oracle restatement of pycocotools'
rleFrPoly, and to_device_batch end to end (vertices in, device batch out, a training step on it)."""
import numpy as np
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu


def _rand_poly(rng, size, n, spread):
    cx, cy = rng.uniform(0, size, 2)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.3, 1.0, n) * spread
    return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).reshape(-1).tolist()


def test_poly2mask_bit_exact_vs_oracle():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(0)
    H, W = 200, 240                                            # non-square, H not a multiple of 32 (column bit ranges straddle words)
    inst = []
    inst.append([[1, 1, 4, 1, 4, 3, 1, 3]])                    # closed-form rectangle
    inst.append([[0, 0, 3, 0, 0, 3]])                          # triangle
    inst.append([[-30.5, -20.2, 60.7, -10.0, 50.1, 300.9, -40.0, 120.3]])      # partly outside on three sides
    inst.append([[10, 10, 100, 100, 100, 10, 10, 100]])        # self-intersecting (bow tie)
    inst.append([_rand_poly(rng, 200, 7, 30), _rand_poly(rng, 200, 5, 25)])    # two polygons, OR-merged
    inst.append([])                                            # instance without polygons -> empty mask
    for _ in range(20):
        inst.append([_rand_poly(rng, 220, rng.randint(3, 40), rng.uniform(2, 90))])
    inst.append([[5.5, 5.5, 5.5, 5.5, 5.5, 5.5]])              # degenerate: a point
    inst.append([(np.array(_rand_poly(rng, 200, 300, 80))).tolist()])          # many vertices
    got = K.poly2mask(inst, H, W).cpu().numpy()
    assert got.shape == (len(inst), H, W) and got.dtype == np.uint8
    for i, polys in enumerate(inst):
        want = R.poly2mask(polys, H, W)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))


def test_poly2mask_full_tile_size():
    from bonai_amd import kernels as K
    rng = np.random.RandomState(1)
    inst = [[_rand_poly(rng, 1024, rng.randint(4, 12), rng.uniform(10, 160))] for _ in range(6)]
    inst.append([[0, 0, 1024, 0, 1024, 1024, 0, 1024]])        # the whole tile
    got = K.poly2mask(inst, 1024, 1024).cpu().numpy()
    for i in (0, 3, 6):
        assert np.array_equal(got[i], R.poly2mask(inst[i], 1024, 1024)), i
    assert got[6].all()


def test_to_device_batch_with_polygons_trains():
    """Samples carry polygon lists (the annotation's `masks` entry, bonai.py:186-199) instead of bitmaps: same device batch as the
    host-rasterised route, and a training step runs on it."""
    import os
    from bonai_amd.config import Config
    from bonai_amd.data import parse_bonai_annotations, to_device_batch
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import synth_bonai_anns
    from oracle.synth_weights import synth_tensor
    size = 256
    anns = synth_bonai_anns(size=size)
    ann = parse_bonai_annotations(dict(width=size, height=size, filename='t.png'), anns)
    rng = np.random.RandomState(2)
    img = rng.randint(0, 255, (size, size, 3)).astype(np.uint8)
    base = dict(img=img, gt_bboxes=ann['bboxes'], gt_labels=ann['labels'], gt_offsets=ann['offsets'])
    host_masks = np.stack([R.poly2mask(m, size, size) for m in ann['masks']])
    b_host = to_device_batch([dict(base, gt_masks=host_masks)])
    b_dev = to_device_batch([dict(base, gt_polygons=ann['masks'])])
    assert torch.equal(b_host['gt_masks'][0], b_dev['gt_masks'][0]) and b_dev['gt_masks'][0].is_cuda
    assert torch.equal(b_host['img'], b_dev['img']) and b_dev['img'].shape == (1, 3, size, size)
    for k in ('gt_bboxes', 'gt_labels', 'gt_offsets'):
        assert torch.equal(b_host[k][0], b_dev[k][0])
    assert b_dev['img_metas'][0]['img_shape'] == (size, size, 3)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(dict(cfg.model, pretrained=None), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    m.load_state_dict({k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()})
    m = m.cuda().train()
    lv = dict(m.train_step(b_dev)['log_vars'].items())
    assert all(np.isfinite(v) for v in lv.values()) and lv['loss_mask'] > 0


def test_flip_of_polygon_samples_equals_bitmap_flip():
    """RandomFlip on a sample that carries polygons -- the device route (rasterise, then mirror the bitmap)
    must give exactly what the host route gives (BitmapMasks.flip of the rasterised masks, structures.py:218-229), for both
    directions and for a double flip; boxes / offsets follow the same rules on both routes."""
    from bonai_amd.data import flip_sample, parse_bonai_annotations, to_device_batch
    from bonai_amd.synth import synth_bonai_anns
    H, W = 192, 256                                            # non-square: a swapped axis cannot pass
    anns = synth_bonai_anns(size=192)
    ann = parse_bonai_annotations(dict(width=W, height=H, filename='t.png'), anns)
    rng = np.random.RandomState(3)
    img = rng.randint(0, 255, (H, W, 3)).astype(np.uint8)
    base = dict(img=img, gt_bboxes=ann['bboxes'], gt_labels=ann['labels'], gt_offsets=ann['offsets'])
    host_masks = np.stack([R.poly2mask(m, H, W) for m in ann['masks']])
    s_host, s_poly = dict(base, gt_masks=host_masks), dict(base, gt_polygons=ann['masks'])
    for dirs in (('horizontal',), ('vertical',), ('horizontal', 'vertical')):
        fh, fp = s_host, s_poly
        for d in dirs:
            fh, fp = flip_sample(fh, d), flip_sample(fp, d)
        assert 'gt_masks' not in fp or fp['gt_masks'] is None
        bh, bp = to_device_batch([fh]), to_device_batch([fp])
        assert torch.equal(bh['gt_masks'][0], bp['gt_masks'][0]), dirs
        assert bp['gt_masks'][0].is_contiguous() and int(bp['gt_masks'][0].sum()) == int(host_masks.sum())
        for k in ('img', ):
            assert torch.equal(bh[k], bp[k])
        for k in ('gt_bboxes', 'gt_offsets'):
            assert torch.equal(bh[k][0], bp[k][0])
        want = host_masks
        for d in dirs:
            want = np.flip(want, axis=2 if d == 'horizontal' else 1)
        assert np.array_equal(bp['gt_masks'][0].cpu().numpy(), want)
    with pytest.raises(KeyError):
        flip_sample(dict(base), 'horizontal')
    with pytest.raises(ValueError):
        flip_sample(s_poly, 'diagonal')


def test_dataset_files_to_training_step(tmp_path):
    """bonai_amd/dataset.py end to end on the device: annotation file + PNG tiles on disk -> BonaiDataset.batches() (filter, parser,
    flip, polygons rasterised and images normalised on the GPU) -> a Trainer step.  The flipped batch's masks equal the host
    pipeline's (rasterise on the host, flip the bitmap), and it is the mirror image of the unflipped batch (masks, image, boxes, offsets)."""
    import json
    import os
    from PIL import Image
    from bonai_amd.config import Config
    from bonai_amd.dataset import BonaiDataset
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import synth_bonai_anns
    size = 1024
    rng = np.random.RandomState(0)
    images, annotations, aid = [], [], 0
    for i in range(2):
        name = f'tile_{i}.png'
        Image.fromarray(rng.randint(0, 255, (size, size, 3)).astype(np.uint8)).save(tmp_path / name)
        images.append(dict(id=10 + i, file_name=name, width=size, height=size))
        for a in synth_bonai_anns(seed=i, size=size):
            aid += 1
            annotations.append(dict(a, id=aid, image_id=10 + i))
    f = tmp_path / 'ann.json'
    json.dump(dict(images=images, annotations=annotations, categories=[dict(id=1, name='building')]), open(f, 'w'))
    per_ratio = {}
    for ratio in (0.0, 1.0):
        dev_ds = BonaiDataset(str(f), str(tmp_path), bbox_type='roof', flip_ratio=ratio, flip_direction='horizontal', seed=1)
        host_ds = BonaiDataset(str(f), str(tmp_path), bbox_type='roof', flip_ratio=ratio, flip_direction='horizontal', seed=1,
                               host_rasteriser=R.poly2mask)
        bd = next(dev_ds.batches(0, 2, shuffle=False))
        bh = next(host_ds.batches(0, 2, shuffle=False))
        for k in ('img',):
            assert torch.equal(bd[k], bh[k])
        for i in range(2):
            assert torch.equal(bd['gt_masks'][i], bh['gt_masks'][i]) and torch.equal(bd['gt_bboxes'][i], bh['gt_bboxes'][i])
            assert torch.equal(bd['gt_offsets'][i], bh['gt_offsets'][i]) and bd['img_metas'][i]['flip'] == (ratio == 1.0)
        per_ratio[ratio] = bd
    for i in range(2):                                          # the flipped batch is the mirror image of the unflipped one
        assert torch.equal(per_ratio[1.0]['gt_masks'][i], per_ratio[0.0]['gt_masks'][i].flip(-1))
        assert torch.equal(per_ratio[1.0]['img'][i], per_ratio[0.0]['img'][i].flip(-1))
        b0, b1 = per_ratio[0.0]['gt_bboxes'][i], per_ratio[1.0]['gt_bboxes'][i]
        assert torch.allclose(b1[:, 0], size - b0[:, 2]) and torch.allclose(b1[:, 2], size - b0[:, 0]) and torch.equal(b1[:, 1], b0[:, 1])
        o0, o1 = per_ratio[0.0]['gt_offsets'][i], per_ratio[1.0]['gt_offsets'][i]
        assert torch.equal(o1[:, 0], -o0[:, 0]) and torch.equal(o1[:, 1], o0[:, 1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    torch.manual_seed(0)
    m = build_detector(dict(cfg.model, pretrained=None), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    tr = Trainer(m, lr=1e-3)
    lv = dict(tr.train_step(bd)['log_vars'].items())
    assert all(np.isfinite(v) for v in lv.values()) and lv['loss_mask'] > 0 and lv['loss_offset'] > 0


def test_prefetching_loader_feeds_the_step(tmp_path):
    """f2: the prefetching loader (decoder threads -> pinned staging ring -> side-stream upload, normalise and
    polygon rasterisation) hands the Trainer the same device batches as the synchronous one, on its own sustains well above the
    step's consumption, and a training loop fed by it runs close to the same loop fed with resident batches and far ahead of
    the loop that decodes synchronously."""
    import json
    import os
    import time
    from PIL import Image
    from bonai_amd.config import Config
    from bonai_amd.dataset import BonaiDataset
    from bonai_amd.engine import Trainer
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import synth_bonai_anns
    size, n_tiles, bs = 1024, 32, 8
    rng = np.random.RandomState(0)
    images, annotations, aid = [], [], 0
    base = (rng.randint(0, 255, (size // 8, size // 8, 3)).astype(np.uint8)).repeat(8, 0).repeat(8, 1)   # compressible, like real tiles
    for i in range(n_tiles):
        name = f'tile_{i}.png'
        Image.fromarray(np.roll(base, 17 * i, axis=1)).save(tmp_path / name, compress_level=3)
        images.append(dict(id=10 + i, file_name=name, width=size, height=size))
        for a in synth_bonai_anns(seed=i, n=80, size=size):
            aid += 1
            annotations.append(dict(a, id=aid, image_id=10 + i))
    f = tmp_path / 'ann.json'
    json.dump(dict(images=images, annotations=annotations, categories=[dict(id=1, name='building')]), open(f, 'w'))
    mk = lambda: BonaiDataset(str(f), str(tmp_path), flip_ratio=0.5, flip_direction='vertical', seed=2)
    a, b = mk(), mk()
    sync = list(a.batches(0, bs, seed=7))
    pre = list(b.batches(0, bs, seed=7, prefetch=2))
    torch.cuda.synchronize()
    assert len(sync) == len(pre) == n_tiles // bs
    for x, y in zip(sync, pre):
        assert torch.equal(x['img'], y['img'])
        for k in ('gt_bboxes', 'gt_labels', 'gt_offsets', 'gt_masks'):
            assert all(torch.equal(p, q) for p, q in zip(x[k], y[k])), k
        assert [m['flip'] for m in x['img_metas']] == [m['flip'] for m in y['img_metas']]
    # loader alone (decoder processes are forked once per dataset, at the first prefetching epoch above: not timed)
    nw = min(16, os.cpu_count() or 8)
    for batch in b.batches(9, bs, seed=7, prefetch=3, workers=nw):
        pass
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    for ep in range(4):
        for batch in b.batches(ep, bs, seed=7, prefetch=3, workers=nw):
            n += batch['img'].shape[0]
    torch.cuda.synchronize()
    rate = n / (time.time() - t0)
    print(f'prefetching loader alone: {rate:.0f} img/s')
    assert rate >= 250, rate
    # fed training loop vs resident batches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    torch.manual_seed(0)
    m = build_detector(dict(cfg.model, pretrained=None), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    tr = Trainer(m, lr=1e-4)
    for batch in sync[:3]:
        tr.train_step(batch)
    torch.cuda.synchronize()

    def run(it):
        torch.cuda.synchronize()
        t = time.time()
        k = 0
        for batch in it:
            tr.train_step(batch)
            k += 1
        torch.cuda.synchronize()
        return (time.time() - t) / k
    resident = min(run(sync * 2) for _ in range(2))
    fed = min(run(batch for ep in range(2) for batch in b.batches(ep, bs, seed=7, prefetch=3, workers=nw))
              for _ in range(2))
    slow = run(batch for batch in a.batches(0, bs, seed=7))          # the synchronous loader in the loop
    print(f'step fed by the prefetching loader {fed * 1e3:.1f} ms, by the synchronous loader {slow * 1e3:.1f} ms, by resident '
          f'batches {resident * 1e3:.1f} ms')
    b.close()
    # resident batches skip what a real-data step has to do somewhere: 8 x 80 polygon rasterisations (here on a side stream, one
    # workgroup with a full LDS bitmap per instance, ~4 ms of GPU time per batch beside the training kernels) and the upload;
    # the prefetcher must hide the HOST side of it (decode + collate: ~45 ms per batch when done in the loop)
    assert fed <= resident * 1.30 + 1e-3, (fed, resident)
    assert fed <= 0.75 * slow, (fed, slow)
