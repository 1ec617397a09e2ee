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


# ------------------------------------------------------------------------------------------------------------------------
# FOR-instance (treeins) final evaluation: two instance predictions (offset grouping, embedding grouping), three classes
# ------------------------------------------------------------------------------------------------------------------------
def _instance_tables_numpy(ps, gs, pi, gi, C):
    """per-instance tables of one (prediction, ground truth) pair of label arrays restricted to the evaluated points:
    sizes, classes (smallest mode of the semantic labels), and for every instance its best IoU with an instance of ITS class
    on the other side (-1 when the other side has no instance of that class)."""
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
    return p_size, p_cls, g_size, g_cls, pairs // ng, pairs % ng, inter


def _instance_tables_device(ps, gs, pi, gi, C):
    """the same tables with everything that touches the points on the GPU (pp_histogram2d, pp_pair_counts)"""
    import torch
    from .. import ops

    def groups(ids, sem):
        m = ids != -1
        u, inv = torch.unique(ids[m], return_inverse=True)
        full = torch.full_like(ids, -1)
        full[m] = inv
        k = int(u.numel())
        if k == 0:
            return full, np.zeros(0, np.int64), np.zeros(0, np.int64)
        table = ops.histogram2d(full, sem, k, C)
        return full, table.sum(1), table.argmax(1)
    pid, p_size, p_cls = groups(pi, ps)
    gid, g_size, g_cls = groups(gi, gs)
    ng = max(len(g_size), 1)
    pa, ga, inter = (t.cpu().numpy() for t in ops.pair_counts(pid, gid, ng))
    return p_size, p_cls, g_size, g_cls, pa, ga, inter


def _best_ious(p_size, p_cls, g_size, g_cls, pa, ga, inter):
    """best IoU of every predicted / ground-truth instance with an instance of its own class on the other side; pairs that
    share no point have IoU 0, so an instance whose class exists on the other side has best >= 0, otherwise -1"""
    pair_iou = inter / (p_size[pa] + g_size[ga] - inter) if len(pa) else np.zeros(0)
    same = p_cls[pa] == g_cls[ga] if len(pa) else np.zeros(0, bool)
    g_has = np.zeros(int(max(p_cls.max(initial=-1), g_cls.max(initial=-1))) + 2, bool)
    g_has[g_cls] = True
    p_has = np.zeros_like(g_has)
    p_has[p_cls] = True
    best_p = np.where(g_has[p_cls], 0.0, -1.0) if len(p_cls) else np.zeros(0)
    best_g = np.where(p_has[g_cls], 0.0, -1.0) if len(g_cls) else np.zeros(0)
    np.maximum.at(best_p, pa[same], pair_iou[same])
    np.maximum.at(best_g, ga[same], pair_iou[same])
    return best_p, best_g


def _treeins_instance_part(tables, iou_list, things, stuff, sem, C, at):
    """one instance prediction of treeins.final_eval (datasets/panoptic/treeins.py:199-424): coverage, precision / recall,
    RQ / SQ / PQ with the reference's conventions -- no "class has ground truth" filtering anywhere, so an absent class gives
    nan (mean of an empty list) or a division by zero exactly where the reference does"""
    p_size, p_cls, g_size, g_cls, pa, ga, inter = tables
    best_p, best_g = _best_ious(p_size, p_cls, g_size, g_cls, pa, ga, inter)
    mucov, mwcov = np.full(C, np.nan), np.full(C, np.nan)
    prec, rec, rq, sq, pq, pqs = (np.zeros(C) for _ in range(6))
    with np.errstate(divide="ignore", invalid="ignore"):
        for c in range(C):
            gm, pm = g_cls == c, p_cls == c
            if gm.any():
                cov = np.maximum(best_g[gm], 0.0)            # ovmax starts at 0 on the ground-truth side
                mucov[c] = cov.sum() / gm.sum()
                mwcov[c] = (cov * g_size[gm]).sum() / g_size[gm].sum()
        for c in things:
            gm, pm = g_cls == c, p_cls == c
            tp_mask = pm & (best_p >= at)                    # ovmax starts at -1: no ground truth of the class -> false positive
            tp, fp = float(tp_mask.sum()), float((pm & ~tp_mask).sum())
            rec[c] = np.float64(tp) / np.float64(gm.sum())
            prec[c] = tp / (tp + fp) if tp + fp > 0 else 0.0
            rq[c] = 2 * prec[c] * rec[c] / (prec[c] + rec[c]) if prec[c] + rec[c] != 0 else 0.0
            sq[c] = best_p[tp_mask].sum() / tp if tp else 0.0
            pq[c] = sq[c] * rq[c]
            pqs[c] = pq[c]
        for c in stuff:
            ok = iou_list[c] >= 0.5
            rq[c], sq[c] = (1.0, iou_list[c]) if ok else (0.0, 0.0)
            pq[c] = sq[c] * rq[c]
            pqs[c] = iou_list[c]
        mp, mr = np.mean(prec[things]), np.mean(rec[things])
        f1 = (2 * mp * mr) / (mp + mr)
    return {"MUCov": mucov[things], "mMUCov": float(np.mean(mucov[things])), "MWCov": mwcov[things],
            "mMWCov": float(np.mean(mwcov[things])), "Precision": prec[things], "mPrecision": float(mp), "Recall": rec[things],
            "mRecall": float(mr), "F1": float(f1), "RQ": rq[sem], "meanRQ": float(np.mean(rq[sem])), "SQ": sq[sem],
            "meanSQ": float(np.mean(sq[sem])), "PQ": pq[sem], "meanPQ": float(np.mean(pq[sem])), "PQStar": pqs[sem],
            "meanPQStar": float(np.mean(pqs[sem])), "PQ_things": pq[things], "meanPQ_things": float(np.mean(pq[things])),
            "meanRQ_things": float(np.mean(rq[things])), "meanSQ_things": float(np.mean(sq[things])),
            "PQ_stuff": pq[stuff], "meanPQ_stuff": float(np.mean(pq[stuff])), "meanRQ_stuff": float(np.mean(rq[stuff])),
            "meanSQ_stuff": float(np.mean(sq[stuff]))}


def panoptic_evaluation_treeins(pre_sem, pre_ins_embed, pre_ins_offset, gt_sem, gt_ins, thing_classes=(1,), stuff_classes=(0,),
                                num_classes=2, iou_threshold=0.5):
    """The FOR-instance final evaluation (torch_points3d/datasets/panoptic/treeins.py:99-497, `final_eval`): semantic
    oAcc / mAcc / IoU / mIoU once, then the instance metrics twice -- for the offset grouping and for the embedding grouping.
    Labels 0-based (ground truth -1 = unclassified), instance id -1 = none.  Differences from the NPM3D form that are kept:
    classes are NOT filtered by "has ground-truth points" (mIoU = sum over ALL classes incl. unclassified / num_classes; a
    class without ground-truth instances gives nan / inf like the reference's numpy arithmetic), coverage starts its maximum
    at 0 and takes the first strictly larger IoU, points enter the instance part when they are not {unclassified, first stuff
    class} in the ground truth or in the prediction.  Device tensors go through the GPU tables (pp_histogram2d, pp_pair_counts).
    Returns {"oAcc", "mAcc", "IoU", "mIoU", "offset": {...}, "embed": {...}}."""
    on_device = hasattr(pre_sem, "is_cuda") and pre_sem.is_cuda
    C = num_classes + 1
    things = np.asarray(thing_classes, np.int64) + 1
    stuff = np.asarray(stuff_classes, np.int64) + 1
    sem = np.arange(1, C)
    if on_device:
        from .. import ops
        ps_all, gs_all = pre_sem.reshape(-1).long() + 1, gt_sem.reshape(-1).long() + 1
        conf = ops.histogram2d(gs_all, ps_all, C, C, allow_skipped=False).astype(np.float64)
        gt_cnt, pr_cnt, tp_cnt = conf.sum(1), conf.sum(0), np.diag(conf).copy()
        keep = ((gs_all != 0) & (gs_all != 1)) | ((ps_all != 0) & (ps_all != 1))
        sel = lambda t: t.reshape(-1).long()[keep]  # noqa: E731
        tables = lambda pi: _instance_tables_device(ps_all[keep], gs_all[keep], sel(pi), sel(gt_ins), C)  # noqa: E731
    else:
        ps_all = np.asarray(pre_sem).reshape(-1).astype(np.int64) + 1
        gs_all = np.asarray(gt_sem).reshape(-1).astype(np.int64) + 1
        gt_cnt = np.bincount(gs_all, minlength=C).astype(np.float64)
        pr_cnt = np.bincount(ps_all, minlength=C).astype(np.float64)
        tp_cnt = np.bincount(gs_all[gs_all == ps_all], minlength=C).astype(np.float64)
        keep = ((gs_all != 0) & (gs_all != 1)) | ((ps_all != 0) & (ps_all != 1))
        sel = lambda a: np.asarray(a).reshape(-1).astype(np.int64)[keep]  # noqa: E731
        tables = lambda pi: _instance_tables_numpy(ps_all[keep], gs_all[keep], sel(pi), sel(gt_ins), C)  # noqa: E731
    with np.errstate(divide="ignore", invalid="ignore"):
        iou_list = tp_cnt / (gt_cnt + pr_cnt - tp_cnt)
        out = {"oAcc": float(tp_cnt.sum() / pr_cnt.sum()), "mAcc": float(np.mean(tp_cnt[sem] / gt_cnt[sem])), "IoU": iou_list,
               "mIoU": float(iou_list.sum() / num_classes)}
    out["offset"] = _treeins_instance_part(tables(pre_ins_offset), iou_list, things, stuff, sem, C, iou_threshold)
    out["embed"] = _treeins_instance_part(tables(pre_ins_embed), iou_list, things, stuff, sem, C, iou_threshold)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# the tracker's per-batch instance metrics (metrics/panoptic_tracker_pointgroup_npm3d.py:678-879)
# ------------------------------------------------------------------------------------------------------------------------
def _clusters_as_labels(clusters, n, device=None):
    """(point -> cluster id, id = position in the list; -1 = in no cluster) for DISJOINT clusters; None when two clusters share
    a point (the tracker is fed the NMS survivors, which may overlap: the general path handles those)"""
    import torch
    lab = torch.full((n,), -1, dtype=torch.int64, device=device)
    total = 0
    for i, c in enumerate(clusters):
        lab[c] = i
        total += int(c.numel())
    return lab if int((lab >= 0).sum()) == total else None


def compute_acc(clusters, predicted_labels, labels, batch, num_instances, iou_threshold):
    """`_compute_acc` (tracker :678-710): share of true / false positive clusters and accuracy.  A cluster is a true positive
    when its best instance IoU reaches the threshold AND the majority class of that ground-truth instance equals the mode of
    the cluster's predicted labels.  `labels` has .instance_labels (1-based per sample, 0 = none), .y, .num_instances.
    One pp_instance_iou launch + two class histograms (pp_histogram2d) instead of a Python loop over clusters."""
    import torch
    from .. import ops
    from ..torch_points_kernels import instance_iou
    dev = batch.device
    n_cl = len(clusters)
    ious = instance_iou(clusters, labels.instance_labels, batch)
    iou_values, gt_ids = ious.max(1)
    offsets = torch.cat((torch.zeros(1, dtype=torch.long, device=dev), num_instances.to(dev).cumsum(-1)))
    n_cls = int(max(int(labels.y.max()), int(predicted_labels.max()))) + 1
    # majority class of every ground-truth instance (global id = offset of its sample + local id - 1), smallest class on ties
    inst = labels.instance_labels.to(dev).long()
    gid = torch.where(inst > 0, offsets[batch.long()] + inst - 1, torch.full_like(inst, -1))
    g_tab = ops.histogram2d(gid, labels.y.to(dev).long(), int(offsets[-1]), n_cls) if int(offsets[-1]) else np.zeros((0, n_cls), np.int64)
    g_cls = g_tab.argmax(1)
    # mode of the predicted labels of every cluster
    cid = torch.cat([torch.full((int(c.numel()),), i, dtype=torch.long, device=dev) for i, c in enumerate(clusters)])
    pts = torch.cat([c.to(dev).long() for c in clusters])
    p_cls = ops.histogram2d(cid, predicted_labels.to(dev).long()[pts], n_cl, n_cls).argmax(1)
    ok = (iou_values >= iou_threshold).cpu().numpy()
    match = g_cls[gt_ids.cpu().numpy()] == p_cls
    tp = int((ok & match).sum())
    fp = int(n_cl - tp)
    total = float(torch.sum(labels.num_instances).cpu().item())
    return tp / total, fp / total, tp / n_cl


def compute_eval(clusters, predicted_labels, labels, batch, num_instances, num_classes, iou_threshold,
                 thing_classes=(2, 3, 4, 6, 7, 8)):
    """`_compute_eval` (tracker :712-879): (cov, wcov, mean precision, mean recall, F1) of one batch.  Predicted instances =
    the clusters (class = mode of the predicted labels), ground-truth instances = every (sample, instance id >= 0) group --
    id 0, the points without an instance, is a group of its own as in the reference -- with the mode of y as class (-1
    skipped); IoUs over the whole batch.  Conventions kept: coverage takes `iou >= ovmax` from 0, a class with ground truth
    but no prediction of that class contributes recall 0 (its ground-truth count is never read), means run over the thing
    classes that have a ground-truth group (an empty set gives nan, like torch.mean of an empty tensor)."""
    import torch
    from .. import ops
    dev = batch.device
    n = int(batch.numel())
    C = int(num_classes)
    pred = predicted_labels.to(dev).long()
    # ground-truth groups: (sample, instance id) with id >= 0
    inst = labels.instance_labels.to(dev).long()
    key = torch.where(inst >= 0, batch.long() * (int(inst.max()) + 2) + inst, torch.full_like(inst, -1))
    gu, ginv = torch.unique(key[key >= 0], return_inverse=True)
    gid = torch.full((n,), -1, dtype=torch.long, device=dev)
    gid[key >= 0] = ginv
    ng = int(gu.numel())
    y = labels.y.to(dev).long()
    g_tab = ops.histogram2d(gid, y + 1, ng, C + 1) if ng else np.zeros((0, C + 1), np.int64)   # column 0 = label -1
    g_size = g_tab.sum(1)
    g_cls = g_tab.argmax(1) - 1                      # torch.mode: the smallest of the most frequent values, -1 included
    # predicted groups may overlap in principle; a cluster list from NMS + painting is disjoint -> label form
    cid = _clusters_as_labels([c.to(dev).long() for c in clusters], n, dev)
    n_cl = len(clusters)
    if cid is None:
        raise ValueError("compute_eval: overlapping clusters (the tracker evaluates the painted instances, which are disjoint)")
    p_tab = ops.histogram2d(cid, pred, n_cl, C) if n_cl else np.zeros((0, C), np.int64)
    p_size, p_cls = p_tab.sum(1), p_tab.argmax(1)
    pa, ga, inter = (t.cpu().numpy() for t in ops.pair_counts(cid, gid, max(ng, 1))) if n_cl and ng else (np.zeros(0, np.int64),) * 3
    keep_g = g_cls >= 0
    pair_iou = inter / (p_size[pa] + g_size[ga] - inter) if len(pa) else np.zeros(0)
    same = (p_cls[pa] == g_cls[ga]) if len(pa) else np.zeros(0, bool)
    best_p, best_g = np.zeros(n_cl), np.zeros(ng)
    np.maximum.at(best_p, pa[same], pair_iou[same])
    np.maximum.at(best_g, ga[same], pair_iou[same])
    mucov, mwcov, prec, rec = (np.zeros(C) for _ in range(4))
    for c in range(C):
        gm, pm = keep_g & (g_cls == c), p_cls == c
        if gm.any() and pm.any():
            mucov[c] = best_g[gm].sum() / gm.sum()
            mwcov[c] = (best_g[gm] * g_size[gm]).sum() / g_size[gm].sum()
    for c in thing_classes:
        gm, pm = keep_g & (g_cls == c), p_cls == c
        if not pm.any():
            continue
        tp = float((pm & (best_p >= iou_threshold) & bool(gm.any())).sum())
        fp = float(pm.sum()) - tp
        rec[c] = tp / gm.sum() if gm.any() else 0.0
        prec[c] = tp / (tp + fp) if tp + fp else 0.0
    have = sorted(set(int(c) for c in thing_classes) & set(int(c) for c in g_cls[keep_g]))
    t = lambda v: torch.tensor(v[have], dtype=torch.float32)  # noqa: E731
    mp, mr = torch.mean(t(prec)), torch.mean(t(rec))
    f1 = torch.tensor(0.) if mp + mr == 0 else (2 * mp * mr) / (mp + mr)
    return torch.mean(t(mucov)), torch.mean(t(mwcov)), mp, mr, f1
