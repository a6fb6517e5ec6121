"""Offset evaluation of the BONAI benchmark: the end-point-error statistics BASELINE.json's metric is quoted in.

Mirrors ``Evaluation.offset_error_vector`` / ``cosine_distance`` of the reference's evaluation script
(tools/bonai/bonai_evaluation.py:241-290): error vector = gt - prediction per matched building, EPE = its length,
aEPE = mean EPE, AE = |atan2(gt) - atan2(pred)|, aAE = mean AE, cosine distance = 1 - cos(angle between the two vectors).
The pairing of predictions with ground-truth buildings is done upstream of this function in the reference by the external
``bstool`` package (mask IoU >= 0.5, not in the tree); ``match_by_iou`` below is that rule on boxes/masks for the fixtures
and tools here.  Host-side numpy on a few thousand 2-vectors: bookkeeping, not part of the accelerated path.
"""
import numpy as np


def cosine_distance(a, b):
    """1 - cos of the angle between row vectors (bonai_evaluation.py:241-257); rows with a zero vector give nan there too."""
    a, b = np.asarray(a, np.float64).reshape(-1, 2), np.asarray(b, np.float64).reshape(-1, 2)
    na, nb = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
    with np.errstate(invalid='ignore', divide='ignore'):
        return 1.0 - (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) / (na * nb)


def offset_error_vector(gt_offsets, pred_offsets):
    """-> dict(aEPE, aAE, cos_distance, max_EPE, EPE[N]) for paired [N,2] offsets (bonai_evaluation.py:259-290)."""
    gt = np.asarray(gt_offsets, np.float64).reshape(-1, 2)
    pr = np.asarray(pred_offsets, np.float64).reshape(-1, 2)
    if gt.shape != pr.shape:
        raise ValueError(f'paired offsets expected, got {gt.shape} and {pr.shape}')
    if gt.shape[0] == 0:
        return dict(aEPE=float('nan'), aAE=float('nan'), cos_distance=float('nan'), max_EPE=float('nan'), EPE=np.zeros(0))
    err = gt - pr
    epe = np.sqrt(err[:, 0] ** 2 + err[:, 1] ** 2)
    ae = np.abs(np.arctan2(gt[:, 1], gt[:, 0]) - np.arctan2(pr[:, 1], pr[:, 0]))
    return dict(aEPE=float(epe.mean()), aAE=float(ae.mean()), cos_distance=float(np.nanmean(cosine_distance(gt, pr))),
                max_EPE=float(epe.max()), EPE=epe)


def match_by_iou(iou, thr=0.5):
    """Greedy one-to-one pairing on an IoU matrix [num_gt, num_pred]: each ground truth takes its best still-free prediction
    with IoU >= thr, ground truths visited in order of their best IoU (highest first).  -> (gt_idx, pred_idx) int arrays."""
    iou = np.asarray(iou, np.float64)
    if iou.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    order = np.argsort(-iou.max(axis=1), kind='stable')
    used = np.zeros(iou.shape[1], bool)
    gi, pi = [], []
    for g in order:
        row = np.where(used, -1.0, iou[g])
        p = int(row.argmax())
        if row[p] >= thr:
            used[p] = True
            gi.append(int(g)); pi.append(p)
    return np.asarray(gi, np.int64), np.asarray(pi, np.int64)
