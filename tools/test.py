#!/usr/bin/env python
"""Entry point with the argv surface of the reference's tools/test.py:17-67 for the LOFT hot path.

    python tools/test.py CONFIG [CHECKPOINT] [--out results.pkl] [--eval] [--ann-file F --img-prefix D] [--num N]

Dataset mode (the annotation file of ``cfg.data.test`` -- or ``--ann-file`` -- exists): every image of the file goes through
`model(return_loss=False, rescale=True, img=[...], img_metas=[[...]])` with samples_per_gpu = 1 (apis/test.py:26,53-72), results
are the reference's 3-tuples (bbox_results, segm_results as COCO RLE, offset_results) and ``--out`` pickles the list exactly as
single_gpu_test returns it.  ``--eval`` evaluates on the spot what the reference evaluates from that pickle with
tools/bonai/bonai_evaluation.py: roof / footprint F1 at IoU 0.5 and the offset aEPE / aAE of the footprint pairs
(bonai_amd/evaluation.py; footprints = roof bitmaps translated by the predicted offset, on the device).
Without dataset files (offline, as in this image) or with ``--synthetic`` the inputs are seeded synthetic tiles.
"""
import argparse
import json
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def run_dataset(model, ds, evaluate=False, eval_kw=None, log=print):
    """-> (results as single_gpu_test returns them, per-image evaluation records or None)."""
    from bonai_amd import evaluation as E
    roi = model.roi_head
    roi.test_cfg['keep_device_masks'] = True
    # simple_test encodes the pasted device bitmaps as RLE itself (one pass): without this it also copied every detection's
    # full-image bool mask to the host (up to 100 x 1 MiB per tile) for a result this loop then threw away
    roi.test_cfg['rle_masks'] = True
    results, records = [], ([] if evaluate else None)
    for i, data in ds.test_batches():
        with torch.no_grad():
            bbox_res, segm, off_res = model(return_loss=False, rescale=True, **data)
        n_det = sum(b.shape[0] for b in bbox_res)
        pasted = roi.last_device_masks if n_det else None
        results.append((bbox_res, segm, off_res))
        if evaluate:
            h, w = data['img_metas'][0][0]['ori_shape'][:2]
            pm = pasted if pasted is not None else torch.zeros(0, h, w, dtype=torch.uint8, device=data['img'][0].device)
            boxes = roi.last_dets if pasted is not None else np.zeros((0, 5), np.float32)
            offs = np.asarray(off_res, np.float32).reshape(-1, 2) if pasted is not None else np.zeros((0, 2), np.float32)
            records.append(E.evaluate_image(pm, boxes, offs, ds.get_ann_info(i), **(eval_kw or {})))
        if (i + 1) % 50 == 0 or i + 1 == len(ds):
            log(f'[{i + 1}/{len(ds)}]')
    return results, records


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('checkpoint', nargs='?')
    ap.add_argument('--out')
    ap.add_argument('--eval', action='store_true', help='roof / footprint F1 and offset aEPE / aAE (tools/bonai/bonai_evaluation.py)')
    ap.add_argument('--ann-file', help='annotation file (default: cfg.data.test.ann_file)')
    ap.add_argument('--img-prefix', help='tile directory (default: cfg.data.test.img_prefix)')
    ap.add_argument('--score-thr', type=float, default=0.4, help='evaluation: detections below are dropped (bonai_evaluation.py:30)')
    ap.add_argument('--min-area', type=float, default=500, help='evaluation: roofs smaller than this many pixels are dropped (:31)')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--num', type=int, default=4, help='synthetic mode: number of tiles')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--bitmap-masks', action='store_true',
                    help='synthetic mode: full-image bool masks like simple_test; default: COCO RLE dicts, i.e. what the reference\'s '
                         'single_gpu_test hands on after encode_mask_results (apis/test.py:59-67), encoded from the device')
    args = ap.parse_args()
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    cfg = Config.fromfile(args.config)
    torch.manual_seed(0)
    if cfg.get('fp16'):                                    # tools/test.py:104-106 wrap_fp16_model
        from bonai_amd import lib as L
        L.set_act16(torch.float16)
    tcfg = (cfg.data.get('test') if cfg.get('data') else None) or {}
    ann = args.ann_file or tcfg.get('ann_file')
    prefix = args.img_prefix if args.img_prefix is not None else tcfg.get('img_prefix', '')
    files = [ann] if isinstance(ann, str) else list(ann or [])
    dataset_mode = not args.synthetic and files and all(os.path.exists(f) for f in files)
    if not dataset_mode and not args.bitmap_masks:
        cfg.test_cfg.rcnn['rle_masks'] = True
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    if args.checkpoint:
        from bonai_amd.checkpoint import load_checkpoint
        load_checkpoint(model, args.checkpoint, strict=True)
    model = model.cuda().eval()
    t0 = time.time()
    if dataset_mode:
        from bonai_amd import evaluation as E
        from bonai_amd.dataset import BonaiDataset
        extra = {k: tcfg[k] for k in ('bbox_type', 'mask_type', 'offset_coordinate', 'resolution', 'classes') if k in tcfg}
        ds = BonaiDataset(ann, prefix, test_mode=True, **extra)
        results, records = run_dataset(model, ds, evaluate=args.eval, eval_kw=dict(score_thr=args.score_thr, min_area=args.min_area))
        torch.cuda.synchronize()
        print(f'{len(ds) / (time.time() - t0):.2f} img/s (decode + inference + mask paste + RLE' + (' + evaluation)' if args.eval else ')'))
        if args.eval:
            print(json.dumps(E.summarize(records), indent=1))
    else:
        if files and not args.synthetic:
            print(f'dataset files of cfg.data.test not found ({files[:1]}...): synthetic tiles', flush=True)
        results = []
        for i in range(args.num):
            data = make_batch(1, args.size, 40, step=i, device='cuda')
            with torch.no_grad():
                res = model(img=[data['img']], img_metas=[data['img_metas']], return_loss=False, rescale=True)
            results.append(res)
            print(f'[{i + 1}/{args.num}] dets={res[0][0].shape[0]} offsets={res[2].shape if hasattr(res[2], "shape") else 0}', flush=True)
        torch.cuda.synchronize()
        print(f'{args.num / (time.time() - t0):.2f} img/s (incl. mask paste + result encoding)')
    if args.out:
        with open(args.out, 'wb') as f:
            pickle.dump(results, f)


if __name__ == '__main__':
    main()
