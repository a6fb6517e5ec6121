"""tests/golden/e2e_256_bf16.npz: the CPU oracle in its 16-bit-points mode (oracle/loft_model_ref.numerics) on the tile,
weights and sampling rule of tests/golden/e2e_256.npz.  TEST INFRASTRUCTURE (build container; `python -m oracle.make_bf16_golden`).

What it pins: the model-level drift of the TIMED bf16 kernels.  The fp32 fixture e2e_256.npz is made by
the reference's own python; this one is made by the restatement that e2e_256.npz pins, with roundings inserted where the HIP path
holds 16-bit data.  So  HIP(bf16) ~ oracle(bf16 points)  at a few 1e-3 (accumulation order only), and
oracle(bf16 points) - oracle(fp32) = the bf16 formulation's own distance from the reference, reported here per quantity."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def run(dtype):
    from bonai_amd.config import Config
    from bonai_amd.loft import build_detector
    from bonai_amd.synth import make_batch
    from oracle import loft_model_ref as M
    from oracle.synth_weights import synth_tensor
    gd = np.load(os.path.join(GOLD, 'e2e_256.npz'))
    size, batch, num_gt = [int(v) for v in gd['meta']]
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'loft_foa', 'loft_foa_r50_fpn_2x_bonai.py'))
    m = build_detector(dict(cfg.model, pretrained=None), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    sd = {k: synth_tensor(k, v.shape) for k, v in m.state_dict().items()}
    cpu = make_batch(batch, size, num_gt)
    torch.set_num_threads(max(1, (os.cpu_count() or 8)))
    with torch.no_grad():
        if dtype is None:
            return M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'],
                                   return_extras=True), (size, batch, num_gt)
        with M.numerics(dtype):
            return M.forward_train(sd, cpu['img'], cpu['gt_bboxes'], cpu['gt_labels'], cpu['gt_masks'], cpu['gt_offsets'],
                                   return_extras=True), (size, batch, num_gt)


def main():
    (l16, e16), meta = run(torch.bfloat16)
    (l32, e32), _ = run(None)
    out = dict(meta=np.array(meta))
    T = lambda t: t.detach().float().cpu().numpy()
    print('quantity                 bf16-points      fp32        rel.diff')
    for k in l16:
        out['log_' + k] = T(l16[k].sum())
        a, b = float(l16[k].sum()), float(l32[k].sum())
        print(f'{k:24s} {a:12.6f} {b:12.6f}  {abs(a - b) / max(1.0, abs(b)):.2e}')
    for i, (f16, f32) in enumerate(zip(e16['feats'], e32['feats'])):
        out[f'feat_{i}_sub'] = T(f16[:, ::8, ::2, ::2])          # (a strided sample; the GPU test runs this mode live for the rest)
        out[f'feat_{i}_l2'] = T(f16.norm())
        print(f'feat_{i} {tuple(f16.shape)}: |bf16pts - fp32| / |fp32| = {float((f16 - f32).norm() / f32.norm()):.3e}')
    n16, n32 = e16['proposals'], e32['proposals']
    for i in range(len(n16)):
        out[f'num_proposals_{i}'] = np.array(n16[i].shape[0])
        out[f'proposals_{i}_top'] = T(n16[i][:64])
    # head outputs on the oracle's OWN sampled RoIs (the RoI lists are part of the fixture: the GPU test feeds them to the heads)
    for k in ('rois', 'pos_rois', 'cls_score', 'bbox_pred', 'offset_pred', 'offset_targets', 'labels'):
        out[k] = T(e16[k])
    out['mask_pred_crop'] = T(e16['mask_pred'][:, :, ::4, ::4])
    out['mask_pred_l2'] = T(e16['mask_pred'].norm())
    same = e16['rois'].shape == e32['rois'].shape and bool((e16['rois'] - e32['rois']).abs().max() < 1e-3)
    print('sampled RoIs identical in both modes:', same, tuple(e16['rois'].shape), tuple(e16['pos_rois'].shape))
    if e16['offset_pred'].shape == e32['offset_pred'].shape:
        d = (e16['offset_pred'] - e32['offset_pred'])
        print(f"offset_pred: |bf16pts - fp32| / |fp32| = {float(d.norm() / e32['offset_pred'].norm()):.3e}")
    np.savez_compressed(os.path.join(GOLD, 'e2e_256_bf16.npz'), **out)
    print('e2e_256_bf16.npz', len(out), 'arrays', round(os.path.getsize(os.path.join(GOLD, 'e2e_256_bf16.npz')) / 1e6, 2), 'MB')


if __name__ == '__main__':
    main()
