"""Panoptic quality of "thing" instances, restating the matching rule of the reference's final evaluation
(torch_points3d/datasets/panoptic/npm3d.py:107-397, the precision / recall / RQ / SQ / PQ part at IoU 0.5):
a predicted instance takes the mode of its predicted semantic labels as class; inside a class it is a true positive
when its best IoU with a ground-truth instance of that class is >= 0.5; per class
precision = TP / #pred, recall = TP / #gt, RQ = 2PR / (P + R), SQ = sum(IoU of TPs) / TP, PQ = SQ * RQ.
Vectorised with a (pred, gt) contingency table instead of the reference's per-pair boolean masks."""
import numpy as np


def _mode_per_group(group, values, n_groups, n_values):
    table = np.zeros((n_groups, n_values), np.int64)
    np.add.at(table, (group, values), 1)
    return table.argmax(1)


def thing_panoptic_quality(pred_sem, pred_ins, gt_sem, gt_ins, thing_classes, iou_threshold=0.5):
    """pred_ins / gt_ins: int arrays, -1 (pred) or <= 0 (gt) = no instance.  Returns dict with per-class and mean PQ."""
    pred_sem, pred_ins = np.asarray(pred_sem).reshape(-1), np.asarray(pred_ins).reshape(-1)
    gt_sem, gt_ins = np.asarray(gt_sem).reshape(-1), np.asarray(gt_ins).reshape(-1)
    n_cls = int(max(pred_sem.max(), gt_sem.max())) + 1
    pm, gm = pred_ins >= 0, gt_ins > 0
    pu, pinv = np.unique(pred_ins[pm], return_inverse=True)
    gu, ginv = np.unique(gt_ins[gm], return_inverse=True)
    pid = np.full(len(pred_ins), -1, np.int64)
    pid[pm] = pinv
    gid = np.full(len(gt_ins), -1, np.int64)
    gid[gm] = ginv
    p_size = np.bincount(pinv, minlength=len(pu))
    g_size = np.bincount(ginv, minlength=len(gu))
    p_cls = _mode_per_group(pinv, pred_sem[pm], len(pu), n_cls) if len(pu) else np.zeros(0, np.int64)
    g_cls = _mode_per_group(ginv, gt_sem[gm], len(gu), n_cls) if len(gu) else np.zeros(0, np.int64)
    both = pm & gm
    pairs, inter = np.unique(pid[both] * max(len(gu), 1) + gid[both], return_counts=True)
    pa, ga = pairs // max(len(gu), 1), pairs % max(len(gu), 1)
    iou = inter / (p_size[pa] + g_size[ga] - inter)
    same = p_cls[pa] == g_cls[ga]
    best = np.zeros(len(pu))
    np.maximum.at(best, pa[same], iou[same])
    out = {"per_class": {}}
    pqs = []
    for c in thing_classes:
        n_pred, n_gt = int((p_cls == c).sum()), int((g_cls == c).sum())
        if n_gt == 0 and n_pred == 0:
            continue
        tp_mask = (p_cls == c) & (best >= iou_threshold)
        tp = int(tp_mask.sum())
        prec = tp / n_pred if n_pred else 0.0
        rec = tp / n_gt if n_gt else 0.0
        rq = 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0
        sq = float(best[tp_mask].sum() / tp) if tp else 0.0
        out["per_class"][int(c)] = {"precision": prec, "recall": rec, "RQ": rq, "SQ": sq, "PQ": sq * rq, "n_gt": n_gt,
                                    "n_pred": n_pred}
        if n_gt:
            pqs.append(sq * rq)
    out["PQ"] = float(np.mean(pqs)) if pqs else 0.0
    return out
