#!/usr/bin/env python
"""Entry point with the argv surface of the reference's tools/train.py:25-64 for the LOFT hot path.

    python tools/train.py configs/loft_foa/loft_foa_r50_fpn_2x_bonai.py [--work-dir D] [--launcher pytorch]
                          [--options k=v ...] [--iters N] [--synthetic]

The BONAI data pipeline is outside the hot-path scope; `--synthetic` (the default, and the
only source available offline) feeds seeded 1024x1024 tiles with the reference's batch-dict keys.  Logging mirrors
TextLoggerHook's key set (default_runtime.py:3-8).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('--work-dir')
    ap.add_argument('--resume-from')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--options', nargs='+', default=[])
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--synthetic', action='store_true', default=True)
    args = ap.parse_args()
    from bonai_amd.config import Config
    from bonai_amd.engine import Trainer, step_lr
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    cfg = Config.fromfile(args.config)
    if args.options:
        cfg.merge_from_dict({k: eval(v) if v.replace('.', '', 1).lstrip('-').isdigit() else v
                             for k, v in (o.split('=', 1) for o in args.options)})
    rank, world = 0, 1
    if args.launcher == 'pytorch':
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', args.local_rank)))
        dist.init_process_group(cfg.dist_params.get('backend', 'nccl'))
    torch.manual_seed(args.seed)
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    if args.resume_from:
        from bonai_amd.checkpoint import load_checkpoint
        load_checkpoint(model, args.resume_from, strict=True)
    tr = Trainer(model, lr=cfg.optimizer.lr, momentum=cfg.optimizer.momentum, weight_decay=cfg.optimizer.weight_decay,
                 max_norm=cfg.optimizer_config.grad_clip.max_norm,
                 loss_scale=(cfg.get('fp16') or {}).get('loss_scale', 1.0))
    bs = cfg.data.get('samples_per_gpu', 8)
    interval = cfg.log_config.get('interval', 10)
    t0 = time.time()
    for it in range(args.iters):
        data = make_batch(bs, 1024, 80, rank=rank, step=it, device='cuda')
        out = tr.train_step(data, lr=step_lr(cfg.optimizer.lr, it, 0, **{k: cfg.lr_config[k] for k in ('warmup_iters', 'warmup_ratio')}))
        if rank == 0 and (it + 1) % interval == 0:
            torch.cuda.synchronize()
            lv = ', '.join(f'{k}: {v:.4f}' for k, v in out['log_vars'].items())
            print(f'Iter [{it + 1}/{args.iters}] time: {(time.time() - t0) / (it + 1):.3f}, {lv}', flush=True)
    if args.work_dir and rank == 0:
        os.makedirs(args.work_dir, exist_ok=True)
        from bonai_amd.checkpoint import save_checkpoint
        save_checkpoint(model, os.path.join(args.work_dir, 'latest.pth'), meta=dict(config=cfg.filename, iter=args.iters))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
