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


def _finish_evaluation(out, iou, have, sem_final, things, stuff, C, p_size, p_cls, g_size, g_cls, best_p, best_g, iou_threshold):
    """per-class coverage / precision / recall / RQ / SQ / PQ from the per-instance tables (shared by the NumPy and the
    device front-end)"""
    mucov, mwcov, prec, rec, rq, sq, pq = (np.zeros(C) for _ in range(7))
    for c in range(C):
        gm, pm = g_cls == c, p_cls == c
        if gm.any() and pm.any():
            mucov[c] = best_g[gm].mean()
            mwcov[c] = (best_g[gm] * g_size[gm]).sum() / g_size[gm].sum()
        if pm.any():
            tp_mask = pm & (best_p >= iou_threshold) & bool(gm.any())
            tp = float(tp_mask.sum())
            prec[c] = tp / pm.sum()
            rec[c] = tp / gm.sum() if gm.any() else 0.0
            rq[c] = 2 * prec[c] * rec[c] / (prec[c] + rec[c]) if prec[c] + rec[c] > 0 else 0.0
            sq[c] = best_p[tp_mask].sum() / tp if tp else 0.0
            pq[c] = sq[c] * rq[c]
    thing_only = np.zeros(C, bool)
    thing_only[things] = True
    prec, rec, rq, sq, pq = (np.where(thing_only, v, 0.0) for v in (prec, rec, rq, sq, pq))
    for c in stuff:
        ok = iou[c] >= iou_threshold
        rq[c], sq[c] = (1.0, iou[c]) if ok else (0.0, 0.0)
        pq[c] = rq[c] * sq[c]
    things_final = [c for c in things if have[c]]
    stuff_final = [c for c in stuff if have[c]]
    mp, mr = float(np.mean(prec[things_final])), float(np.mean(rec[things_final]))
    out.update({
        "MUCov": mucov[things], "mMUCov": float(np.mean(mucov[things_final])), "MWCov": mwcov[things],
        "mMWCov": float(np.mean(mwcov[things_final])), "Precision": prec[things], "mPrecision": mp, "Recall": rec[things],
        "mRecall": mr, "F1": 2 * mp * mr / (mp + mr) if mp + mr > 0 else 0.0,
        "RQ": rq[1:], "SQ": sq[1:], "PQ": pq[1:], "meanRQ": float(np.mean(rq[sem_final])), "meanSQ": float(np.mean(sq[sem_final])),
        "meanPQ": float(np.mean(pq[sem_final])), "PQ_things": pq[things], "meanRQ_things": float(np.mean(rq[things_final])),
        "meanSQ_things": float(np.mean(sq[things_final])), "meanPQ_things": float(np.mean(pq[things_final])),
        "PQ_stuff": pq[stuff], "meanPQ_stuff": float(np.mean(pq[stuff_final])) if stuff_final else 0.0})
    return out


def panoptic_evaluation(pred_sem, pred_ins, gt_sem, gt_ins, thing_classes=(2, 3, 4, 6, 7, 8), stuff_classes=(0, 1, 5),
                        num_classes=9, iou_threshold=0.5):
    """The reference's final scene evaluation (torch_points3d/datasets/panoptic/npm3d.py:107-397, `final_eval`) restated
    with one (prediction, ground-truth) contingency table instead of per-pair boolean masks and per-point loops.
    Labels are 0-based like the reference's inputs (ground truth -1 = unlabelled); instance id -1 = none.
    Quirks kept on purpose: classes are those with ground-truth POINTS; only points that are thing in the ground truth
    or in the prediction enter the instance part; an instance's class is the (smallest) mode of its semantic labels;
    a prediction is a true positive when its best IoU with a ground-truth instance OF ITS CLASS reaches the threshold
    (no one-to-one matching); the mIoU numerator also carries class "unlabelled" when it occurs.
    Returns a dict of the quantities the reference logs (means over the classes present in the ground truth)."""
    pred_sem = np.asarray(pred_sem).reshape(-1).astype(np.int64) + 1
    gt_sem = np.asarray(gt_sem).reshape(-1).astype(np.int64) + 1
    pred_ins = np.asarray(pred_ins).reshape(-1).astype(np.int64)
    gt_ins = np.asarray(gt_ins).reshape(-1).astype(np.int64)
    C = num_classes + 1
    things = np.asarray(thing_classes, np.int64) + 1
    stuff = np.asarray(stuff_classes, np.int64) + 1
    # ---- semantic part
    gt_cnt = np.bincount(gt_sem, minlength=C).astype(np.float64)
    pr_cnt = np.bincount(pred_sem, minlength=C).astype(np.float64)
    tp_cnt = np.bincount(gt_sem[gt_sem == pred_sem], minlength=C).astype(np.float64)
    have = gt_cnt > 0
    iou = np.where(have, tp_cnt / np.maximum(gt_cnt + pr_cnt - tp_cnt, 1), 0.0)
    sem_final = [c for c in range(1, C) if have[c]]
    out = {"oAcc": tp_cnt.sum() / pr_cnt.sum(), "mAcc": float(np.mean(tp_cnt[sem_final] / gt_cnt[sem_final])),
           "IoU": iou, "mIoU": float(iou.sum() / len(sem_final))}
    # ---- instances: only points that are thing in gt or prediction
    is_thing = np.zeros(C, bool)
    is_thing[things] = True
    keep = is_thing[gt_sem] | is_thing[pred_sem]
    ps, gs, pi, gi = pred_sem[keep], gt_sem[keep], pred_ins[keep], gt_ins[keep]

    def groups(ids, sem):
        m = ids != -1
        u, inv = np.unique(ids[m], return_inverse=True)
        full = np.full(len(ids), -1, np.int64)
        full[m] = inv
        size = np.bincount(inv, minlength=len(u))
        cls = _mode_per_group(inv, sem[m], len(u), C) if len(u) else np.zeros(0, np.int64)
        return full, size, cls
    pid, p_size, p_cls = groups(pi, ps)
    gid, g_size, g_cls = groups(gi, gs)
    both = (pid >= 0) & (gid >= 0)
    ng = max(len(g_size), 1)
    pairs, inter = np.unique(pid[both] * ng + gid[both], return_counts=True)
    pa, ga = pairs // ng, pairs % ng
    pair_iou = inter / (p_size[pa] + g_size[ga] - inter)
    same = p_cls[pa] == g_cls[ga]
    best_p = np.zeros(len(p_size))
    np.maximum.at(best_p, pa[same], pair_iou[same])
    best_g = np.zeros(len(g_size))
    np.maximum.at(best_g, ga[same], pair_iou[same])
    return _finish_evaluation(out, iou, have, sem_final, things, stuff, C, p_size, p_cls, g_size, g_cls, best_p, best_g,
                              iou_threshold)


def panoptic_evaluation_device(pred_sem, pred_ins, gt_sem, gt_ins, thing_classes=(2, 3, 4, 6, 7, 8), stuff_classes=(0, 1, 5),
                               num_classes=9, iou_threshold=0.5):
    """`panoptic_evaluation` for device tensors: everything that touches the N points runs on the GPU -- the class
    confusion matrix and the instance x class tables through pp_histogram2d, the (prediction, ground truth) instance
    contingency table through pp_pair_counts (csrc/pp_eval.hip); what is left on the host are tables with one row per
    instance or class.  Same quantities, same quirks, same results as the NumPy version above."""
    import torch
    from .. import ops
    C = num_classes + 1
    ps_all = pred_sem.reshape(-1).long() + 1
    gs_all = gt_sem.reshape(-1).long() + 1
    pi_all, gi_all = pred_ins.reshape(-1).long(), gt_ins.reshape(-1).long()
    dev = ps_all.device
    things = np.asarray(thing_classes, np.int64) + 1
    stuff = np.asarray(stuff_classes, np.int64) + 1
    # ---- semantic part: one confusion matrix (ground truth x prediction)
    conf = ops.histogram2d(gs_all, ps_all, C, C, allow_skipped=False).astype(np.float64)  # raises on labels < -1 or >= num_classes
    gt_cnt, pr_cnt, tp_cnt = conf.sum(1), conf.sum(0), np.diag(conf).copy()
    have = gt_cnt > 0
    iou = np.where(have, tp_cnt / np.maximum(gt_cnt + pr_cnt - tp_cnt, 1), 0.0)
    sem_final = [c for c in range(1, C) if have[c]]
    out = {"oAcc": tp_cnt.sum() / pr_cnt.sum(), "mAcc": float(np.mean(tp_cnt[sem_final] / gt_cnt[sem_final])),
           "IoU": iou, "mIoU": float(iou.sum() / len(sem_final))}
    # ---- instances: only points that are thing in gt or prediction
    is_thing = torch.zeros(C, dtype=torch.bool, device=dev)
    is_thing[torch.from_numpy(things).to(dev)] = True
    keep = is_thing[gs_all] | is_thing[ps_all]
    ps, gs, pi, gi = ps_all[keep], gs_all[keep], pi_all[keep], gi_all[keep]

    def groups(ids, sem):
        m = ids != -1
        u, inv = torch.unique(ids[m], return_inverse=True)
        full = torch.full_like(ids, -1)
        full[m] = inv
        k = int(u.numel())
        if k == 0:
            return full, np.zeros(0, np.int64), np.zeros(0, np.int64)
        table = ops.histogram2d(full, sem, k, C)      # rows with full == -1 are skipped by the kernel
        return full, table.sum(1), table.argmax(1)
    pid, p_size, p_cls = groups(pi, ps)
    gid, g_size, g_cls = groups(gi, gs)
    ng = max(len(g_size), 1)
    pa, ga, inter = (t.cpu().numpy() for t in ops.pair_counts(pid, gid, ng))
    pair_iou = inter / (p_size[pa] + g_size[ga] - inter) if len(pa) else np.zeros(0)
    same = p_cls[pa] == g_cls[ga] if len(pa) else np.zeros(0, bool)
    best_p = np.zeros(len(p_size))
    np.maximum.at(best_p, pa[same], pair_iou[same])
    best_g = np.zeros(len(g_size))
    np.maximum.at(best_g, ga[same], pair_iou[same])
    return _finish_evaluation(out, iou, have, sem_final, things, stuff, C, p_size, p_cls, g_size, g_cls, best_p, best_g,
                              iou_threshold)
