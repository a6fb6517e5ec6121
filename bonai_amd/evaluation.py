"""Evaluation of the BONAI benchmark: offset end-point-error statistics (BASELINE.json's metric) and roof / footprint F1.

Mirrors ``Evaluation.offset_error_vector`` / ``cosine_distance`` of the reference's evaluation script
(tools/bonai/bonai_evaluation.py:241-290): error vector = gt - prediction per matched building, EPE = its length,
aEPE = mean EPE, AE = |atan2(gt) - atan2(pred)|, aAE = mean AE, cosine distance = 1 - cos(angle between the two vectors).
The pairing of predictions with ground-truth buildings (``pair_by_iou``) is the rule of the reference's own
``get_confusion_matrix_indexes`` (:461-475); ``match_by_iou`` is a one-to-one variant used by bench.py's model-vs-reference
comparison.  What the external ``bstool`` package does upstream of the pairing (score / area filter, footprint = roof moved by
-offset) is restated on bitmaps in ``evaluate_image``.  The bitmaps stay on the device (footprints: kernels.mask_translate,
ground truth: kernels.poly2mask, intersections: windowed AND + count); the pairing itself is host-side numpy on a few hundred
numbers per image: bookkeeping, not part of the accelerated path.
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


# ---- dataset-level evaluation (round 4) ----------------------------------------------------------------------
# The reference evaluates a results pickle in three steps (tools/bonai/bonai_evaluation.py): (1) the external ``bstool`` package
# turns every (bbox, roof mask, offset) detection with score >= 0.4 and area >= 500 px (:29-31, :64-91) into a roof polygon and a
# FOOTPRINT polygon = roof translated by -offset (offset_model='footprint2roof'); (2) predictions and ground truth of one image
# are paired by ``iou = inter / (area_pred + area_gt - inter + 1.0) >= 0.5`` -- ALL pairs over the threshold, not a one-to-one
# assignment (:461-475); TP = number of pairs, FN / FP = ground truths / predictions in no pair, F1 from those (:375-389);
# (3) offsets of the pairs found on FOOTPRINTS give aEPE / aAE (:259-290).  Steps (2) and (3) are in the tree and are restated
# here exactly; step (1) is bstool's (absent: "parity unpinned" for the polygon <-> bitmap difference) and is restated on the
# bitmaps simple_test returns: areas and intersections are pixel counts, the footprint is kernels.mask_translate of the roof.

def pair_by_iou(inter, area_pred, area_gt, thr=0.5):
    """bonai_evaluation.py:461-475 on an intersection matrix [num_pred, num_gt] -> dict(pred_TP, gt_TP, gt_FN, pred_FP, iou)."""
    inter = np.asarray(inter, np.float64).reshape(len(area_pred), len(area_gt))
    union = np.asarray(area_pred, np.float64)[:, None] + np.asarray(area_gt, np.float64)[None, :]
    iou = inter / (union - inter + 1.0)
    idx = np.argwhere(iou >= thr)
    pred_tp, gt_tp = idx[:, 0].tolist(), idx[:, 1].tolist()
    return dict(pred_TP=pred_tp, gt_TP=gt_tp, gt_FN=sorted(set(range(len(area_gt))) - set(gt_tp)),
                pred_FP=sorted(set(range(len(area_pred))) - set(pred_tp)), iou=iou)


def f1_scores(tp, fn, fp):
    """bonai_evaluation.py:375-389 (the reference divides by zero on an empty confusion matrix; nan here)."""
    prec = tp / (tp + fp) if tp + fp else float('nan')
    rec = tp / (tp + fn) if tp + fn else float('nan')
    f1 = 2 * prec * rec / (prec + rec) if tp else (0.0 if (tp + fp and tp + fn) else float('nan'))
    return dict(F1_score=f1, Precision=prec, Recall=rec, TP=int(tp), FN=int(fn), FP=int(fp))


def _intersections(pm, gm, boxes):
    """pm uint8 [P,H,W], gm uint8 [G,H,W] on the device, boxes: host int [P,4] windows that contain each prediction's pixels
    -> int64 [P,G] pixel intersections (one windowed AND + count per prediction; no [P, G, H, W] tensor)."""
    import torch
    P, G = pm.shape[0], gm.shape[0]
    out = torch.zeros(P, G, dtype=torch.int64, device=pm.device)
    H, W = pm.shape[1:]
    for i in range(P):
        x0, y0, x1, y1 = (int(v) for v in boxes[i])
        x0, y0, x1, y1 = max(0, x0), max(0, y0), min(W, x1), min(H, y1)
        if x1 > x0 and y1 > y0 and G:
            out[i] = (gm[:, y0:y1, x0:x1] & pm[i:i + 1, y0:y1, x0:x1]).flatten(1).sum(1)
    return out


def evaluate_image(pred_masks, pred_boxes, pred_offsets, ann, score_thr=0.4, min_area=500, iou_thr=0.5):
    """One image.  pred_masks: device uint8 [P,H,W] roof bitmaps (simple_test's pasted masks), pred_boxes: host [P,5]
    (x1, y1, x2, y2, score), pred_offsets: host [P,2], ann: BonaiDataset.get_ann_info (roof_masks / footprint_masks polygon
    lists, offsets).  -> dict(roof=pairing, footprint=pairing, gt_offsets, pred_offsets) with the reference's pairing lists."""
    import torch
    from . import kernels as K
    H, W = pred_masks.shape[1:] if pred_masks.dim() == 3 else (0, 0)
    pb = np.asarray(pred_boxes, np.float32).reshape(-1, 5)
    po = np.asarray(pred_offsets, np.float32).reshape(-1, 2)
    dev = pred_masks.device
    area = pred_masks.flatten(1).sum(1).cpu().numpy() if pb.shape[0] else np.zeros(0)
    keep = np.where((pb[:, 4] >= score_thr) & (area >= min_area))[0]
    pm = pred_masks[torch.as_tensor(keep, device=dev)].contiguous() if keep.size else pred_masks[:0]
    pb, po = pb[keep], po[keep]
    fp = K.mask_translate(pm, torch.as_tensor(po, device=dev)) if keep.size else pm
    n_gt = len(ann['roof_masks'])
    g_roof = K.poly2mask(ann['roof_masks'], H, W, device=dev) if n_gt else pm[:0]
    g_fp = K.poly2mask(ann['footprint_masks'], H, W, device=dev) if n_gt else pm[:0]
    # windows: mask_paste paints inside [floor(x1) - 1, ceil(x2) + 1); the footprint is that window moved by -round(offset)
    win = np.stack([np.floor(pb[:, 0]) - 2, np.floor(pb[:, 1]) - 2, np.ceil(pb[:, 2]) + 2, np.ceil(pb[:, 3]) + 2], 1) if keep.size \
        else np.zeros((0, 4))
    sh = np.sign(po) * np.floor(np.abs(po) + 0.5)
    win_fp = win - np.concatenate([sh, sh], 1) if keep.size else win
    out = {}
    for name, p_, g_, w_ in (('roof', pm, g_roof, win), ('footprint', fp, g_fp, win_fp)):
        inter = _intersections(p_, g_, w_).cpu().numpy()
        ap = p_.flatten(1).sum(1).cpu().numpy() if p_.shape[0] else np.zeros(0)
        ag = g_.flatten(1).sum(1).cpu().numpy() if g_.shape[0] else np.zeros(0)
        out[name] = pair_by_iou(inter, ap, ag, iou_thr)
    gt_off = np.asarray(ann['offsets'], np.float32).reshape(-1, 2)
    out['gt_offsets'] = gt_off[out['footprint']['gt_TP']]
    out['pred_offsets'] = po[out['footprint']['pred_TP']]
    out['num_pred'], out['num_gt'] = int(keep.size), int(n_gt)
    return out


def summarize(per_image):
    """Dataset totals from evaluate_image results: roof / footprint F1 (bonai_evaluation.py:350-396) and the offset error
    vector statistics of the footprint pairs (:259-290)."""
    res = {}
    for name in ('roof', 'footprint'):
        tp = sum(len(r[name]['gt_TP']) for r in per_image)
        fn = sum(len(r[name]['gt_FN']) for r in per_image)
        fp = sum(len(r[name]['pred_FP']) for r in per_image)
        res[name] = f1_scores(tp, fn, fp)
    gt = np.concatenate([r['gt_offsets'] for r in per_image]) if per_image else np.zeros((0, 2))
    pr = np.concatenate([r['pred_offsets'] for r in per_image]) if per_image else np.zeros((0, 2))
    off = offset_error_vector(gt, pr)
    res['offset'] = dict(aEPE=off['aEPE'], aAE=off['aAE'], cos_distance=off['cos_distance'], max_EPE=off['max_EPE'], pairs=int(gt.shape[0]))
    return res
