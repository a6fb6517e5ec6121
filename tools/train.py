#!/usr/bin/env python
"""Entry point with the argv surface of the reference's tools/train.py:25-64 for the LOFT hot path.

    python tools/train.py configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py [--work-dir D] [--launcher pytorch]
                          [--options k=v ...] [--iters N] [--synthetic]

Data: when the annotation files of ``cfg.data.train`` exist, batches come from them (bonai_amd/dataset.py: the reference's
BONAI dataset + train pipeline semantics, polygons rasterised and images normalised on the device), sharded over the ranks like
DistributedGroupSampler; otherwise -- offline, as in this image -- or with ``--synthetic``, seeded 1024x1024 tiles with the
reference's batch-dict keys.  Logging mirrors TextLoggerHook's key set (default_runtime.py:3-8).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse_option(opt):
    """'key=value' -> (key, value): Python literals (numbers, tuples, lists, True/False/None, quoted strings) through
    ast.literal_eval -- never eval() --, anything else stays the string it is (mmcv DictAction's behaviour for plain words)."""
    import ast
    if '=' not in opt:
        raise ValueError(f'--options takes key=value pairs, got {opt!r}')
    k, v = opt.split('=', 1)
    try:
        return k, ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return k, v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('--work-dir')
    ap.add_argument('--resume-from', help='checkpoint written by this tool / the reference: weights, SGD momentum, iter, epoch')
    ap.add_argument('--load-from', help='weights only (apis/train.py:141-142)')
    ap.add_argument('--pretrained', help='local backbone checkpoint (torchvision / model-zoo keys) replacing cfg.model.pretrained')
    ap.add_argument('--iters-per-epoch', type=int, default=0, help='synthetic stream: iterations that count as one epoch (0: one epoch)')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--options', nargs='+', default=[])
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--synthetic', action='store_true', help='seeded synthetic tiles even when the dataset files are present')
    ap.add_argument('--prefetch', type=int, default=2, help='dataset batches decoded / uploaded ahead of the step (0: synchronous loader)')
    ap.add_argument('--graph', action='store_true', help='backbone + neck forward / backward as two hipGraphs (bonai_amd/graphs.py)')
    args = ap.parse_args()
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer, step_lr
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    cfg = Config.fromfile(args.config)
    if args.options:
        cfg.merge_from_dict(dict(parse_option(o) for o in args.options))
    rank, world = 0, 1
    if args.launcher == 'pytorch':
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', args.local_rank)))
        dist.init_process_group(cfg.dist_params.get('backend', 'nccl'))
    torch.manual_seed(args.seed)
    from bonai_amd.checkpoint import load_checkpoint, save_checkpoint
    if cfg.get('fp16'):                                    # Fp16OptimizerHook recipe: half activations, fp32 masters, static scale
        from bonai_amd import lib as L
        L.set_act16(torch.float16)
    if args.pretrained:
        cfg.model['pretrained'] = args.pretrained
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    if args.load_from:
        load_checkpoint(model, args.load_from, strict=False)
    tr = Trainer(model, lr=cfg.optimizer.lr, momentum=cfg.optimizer.momentum, weight_decay=cfg.optimizer.weight_decay,
                 max_norm=cfg.optimizer_config.grad_clip.max_norm,
                 loss_scale=(cfg.get('fp16') or {}).get('loss_scale', 1.0), graph_features=args.graph)
    start_iter = 0
    if args.resume_from:                                   # mmcv runner.resume: weights + optimizer state + iter / epoch
        ckpt = load_checkpoint(model, args.resume_from, strict=True)
        if ckpt.get('optimizer') is not None:
            tr.load_optimizer_state(ckpt['optimizer'])
        start_iter = int(ckpt.get('meta', {}).get('iter', 0))
    bs = cfg.data.get('samples_per_gpu', 8)
    interval = cfg.log_config.get('interval', 10)
    ipe = args.iters_per_epoch or max(args.iters, 1)
    sched = {k: cfg.lr_config[k] for k in ('warmup_iters', 'warmup_ratio', 'step') if k in cfg.lr_config}
    dataset = None
    tcfg = cfg.data.get('train') if cfg.get('data') else None
    if tcfg is not None and not args.synthetic:
        files = [tcfg['ann_file']] if isinstance(tcfg['ann_file'], str) else list(tcfg['ann_file'])
        if files and all(os.path.exists(f) for f in files):
            from bonai_amd.dataset import BonaiDataset
            flip = next((p for p in tcfg.get('pipeline', []) if p.get('type') == 'RandomFlip'), {})
            extra = {k: tcfg[k] for k in ('offset_coordinate', 'resolution', 'ignore_buildings', 'filter_empty_gt', 'classes')
                     if k in tcfg}                     # (bonai.py:18-35: the dataset's own keyword arguments)
            dataset = BonaiDataset(tcfg['ann_file'], tcfg.get('img_prefix', ''), bbox_type=tcfg.get('bbox_type', 'roof'),
                                   mask_type=tcfg.get('mask_type', 'roof'), flip_ratio=flip.get('flip_ratio', 0.0) or 0.0,
                                   flip_direction=flip.get('direction', 'horizontal'), seed=args.seed + rank, **extra)
            ipe = args.iters_per_epoch or max(1, len(dataset.epoch_indices(0, bs, rank, world)) // bs)
        elif rank == 0:
            print(f'dataset files of cfg.data.train not found ({files[:1]}...): synthetic tiles', flush=True)

    def stream():
        if dataset is None:
            for it in range(start_iter, args.iters):
                yield it, make_batch(bs, 1024, 80, rank=rank, step=it, device='cuda')
            return
        it = start_iter
        while it < args.iters:
            for data in dataset.batches(it // ipe, bs, rank, world, seed=args.seed, prefetch=args.prefetch,
                                        workers=cfg.data.get('workers_per_gpu', 2) * 4):
                if it >= args.iters:
                    return
                yield it, data
                it += 1
    t0 = time.time()
    for it, data in stream():
        out = tr.train_step(data, lr=step_lr(cfg.optimizer.lr, it, it // ipe, **sched))
        if rank == 0 and (it + 1) % interval == 0:
            torch.cuda.synchronize()
            lv = ', '.join(f'{k}: {v:.4f}' for k, v in out['log_vars'].items())
            print(f'Epoch [{it // ipe + 1}][{it % ipe + 1}/{ipe}] time: {(time.time() - t0) / (it - start_iter + 1):.3f}, {lv}', flush=True)
    if args.work_dir and rank == 0:
        os.makedirs(args.work_dir, exist_ok=True)
        save_checkpoint(model, os.path.join(args.work_dir, 'latest.pth'), optimizer_state=tr.optimizer_state_dict(),
                        meta=dict(config=cfg.filename, iter=args.iters, epoch=args.iters // ipe))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
