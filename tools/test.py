#!/usr/bin/env python
"""Entry point with the argv surface of the reference's tools/test.py:17-67 for the LOFT hot path.

    python tools/test.py configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py [CHECKPOINT] [--out results.pkl] [--num 4]

Runs `model(return_loss=False, rescale=True, img=[...], img_metas=[[...]])` (apis/test.py:26, samples_per_gpu = 1) and
collects the reference's result 3-tuples (bbox_results, segm_results, offset_results).  The BONAI dataset/evaluator are
outside the hot-path scope; offline the inputs are seeded synthetic tiles.
"""
import argparse
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('checkpoint', nargs='?')
    ap.add_argument('--out')
    ap.add_argument('--num', type=int, default=4)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--bitmap-masks', action='store_true',
                    help='return full-image bool masks like simple_test; default: COCO RLE dicts, i.e. what the reference\'s '
                         'single_gpu_test hands on after encode_mask_results (apis/test.py:59-67), encoded from the device')
    args = ap.parse_args()
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    cfg = Config.fromfile(args.config)
    if not args.bitmap_masks:
        cfg.test_cfg.rcnn['rle_masks'] = True
    torch.manual_seed(0)
    if cfg.get('fp16'):                                    # tools/test.py:104-106 wrap_fp16_model
        from bonai_amd import lib as L
        L.set_act16(torch.float16)
    model = build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    if args.checkpoint:
        from bonai_amd.checkpoint import load_checkpoint
        load_checkpoint(model, args.checkpoint, strict=True)
    model = model.cuda().eval()
    results = []
    t0 = time.time()
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
