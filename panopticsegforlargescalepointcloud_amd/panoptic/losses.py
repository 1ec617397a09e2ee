"""Panoptic training losses with the reference's semantics (torch_points3d/core/losses/panoptic_losses.py):
offset_loss :7-23, instance_ious :25-90 (non-mask branch), instance_iou_loss :92-114,
discriminative_loss(_single) :203-343 (L1-norm variant, delta_v 0.5, delta_d 1.5, reg 0.001, mean over batch elements).
Pure tensor algebra; segment sums go through the MI355X scatter kernel.  Pinned against the reference's own
implementation by tests/golden/loss_cases.npz."""
import torch

from ..torch_points_kernels import instance_iou, instance_iou_csr
from ..torch_scatter import gather, scatter


def semantic_nll(log_probs, labels, ignore_index):
    """mean over the non-ignored rows of -log_probs[i, labels[i]] == torch.nn.functional.nll_loss(..., ignore_index) (reference
    pointgroup3heads.py:367) written as gather + masked mean: torch's nll_loss forward reduces in ONE workgroup (0.33 ms at
    325 k rows)"""
    keep = labels != ignore_index
    picked = log_probs.gather(1, labels.clamp_min(0).unsqueeze(1)).squeeze(1)
    return -(picked * keep).sum() / keep.sum()


def offset_loss(pred_offsets, gt_offsets, total_instance_points):
    """L1 regression + direction (negative cosine) terms over the instance points, both normalised by the number of
    instance points (panoptic_losses.py:7-23)."""
    denom = total_instance_points + 1e-6
    l1 = (pred_offsets - gt_offsets).abs().sum(-1)

    def unit(v):
        return v / (v.norm(p=2, dim=1, keepdim=True) + 1e-8)

    cosine = (unit(gt_offsets) * unit(pred_offsets)).sum(-1)
    return {"offset_norm_loss": l1.sum() / denom, "offset_dir_loss": (-cosine).sum() / denom}


def _proposal_rows(predicted_clusters, clusters_csr, device):
    """(points int64 [R], proposal id int64 [R], n_proposals): the proposals' rows concatenated in proposal order"""
    if clusters_csr is not None:
        sizes = clusters_csr.sizes().long()
        return clusters_csr.points.long(), torch.repeat_interleave(torch.arange(clusters_csr.n, device=device), sizes), clusters_csr.n
    pts = torch.cat([c.to(device) for c in predicted_clusters]).long()
    pid = torch.cat([torch.full((len(c),), i, dtype=torch.int64, device=device) for i, c in enumerate(predicted_clusters)])
    return pts, pid, len(predicted_clusters)


def _gt_layout(instance_labels, batch):
    """per batch element: number of ground-truth instances (= its largest id) and the offset of its first column"""
    nb = int(batch[-1]) + 1
    k = torch.zeros(nb, dtype=torch.int64, device=batch.device).scatter_reduce_(0, batch.long(), instance_labels.long(), "amax")
    off = torch.cumsum(k, 0) - k
    return k, off, int(k.sum())


def mask_instance_ious(predicted_clusters, instance_labels, batch, mask_scores_sigmoid, clusters_csr=None):
    """IoU of every proposal's MASKED points (mask score > 0.5) with every ground-truth instance of its batch element
    (panoptic_losses.py:35-88): intersection / (masked proposal points + instance size - intersection + 1e-5); columns of
    other batch elements stay 0.  One scatter instead of the reference's proposals x instances Python loop."""
    dev = instance_labels.device
    if batch is None:
        batch = torch.zeros_like(instance_labels)
    pts, pid, n_prop = _proposal_rows(predicted_clusters, clusters_csr, dev)
    k, off, total = _gt_layout(instance_labels, batch)
    inst, b = instance_labels.long(), batch.long()
    col_all = off[b] + inst - 1                                   # column of every point's own instance (inst >= 1)
    gt_size = torch.zeros(total, dtype=torch.float32, device=dev).index_add_(0, col_all[inst > 0], torch.ones(int((inst > 0).sum()), device=dev))
    keep = mask_scores_sigmoid.reshape(-1) > 0.5
    prop_total = torch.zeros(n_prop, dtype=torch.float32, device=dev).index_add_(0, pid[keep], torch.ones(int(keep.sum()), device=dev))
    hit = keep & (inst[pts] > 0)
    inter = torch.zeros(n_prop * max(total, 1), dtype=torch.float32, device=dev)
    inter.index_add_(0, pid[hit] * total + col_all[pts[hit]], torch.ones(int(hit.sum()), device=dev))
    inter = inter[: n_prop * total].view(n_prop, total)
    # the reference fills only the columns of the proposal's own batch element (of its first point)
    sample = b[pts[torch.cat([torch.zeros(1, dtype=torch.bool, device=dev), pid[1:] == pid[:-1]]).logical_not()]]
    cols = torch.arange(total, device=dev)
    own = (cols[None, :] >= off[sample][:, None]) & (cols[None, :] < (off[sample] + k[sample])[:, None])
    ious = inter / (prop_total[:, None] + gt_size[None, :] - inter + 1e-5)
    return torch.where(own, ious, torch.zeros_like(ious))


def instance_ious(predicted_clusters, cluster_scores, instance_labels, batch, mask_scores_sigmoid=None,
                  cal_iou_based_on_mask=False, clusters_csr=None):
    if cal_iou_based_on_mask:
        assert mask_scores_sigmoid is not None
        return mask_instance_ious(predicted_clusters, instance_labels, batch, mask_scores_sigmoid, clusters_csr)
    if clusters_csr is not None:
        return instance_iou_csr(clusters_csr, instance_labels, batch)
    return instance_iou(predicted_clusters, instance_labels, batch)


def mask_loss(ious, predicted_clusters, mask_scores_sigmoid, instance_labels, batch, clusters_csr=None):
    """BCE of the per-point mask scores of the proposals whose best IoU exceeds 0.5: target 1 on the points of that
    best-matching ground-truth instance, 0 on the proposal's other points; points of all other proposals get weight 0
    (their target value 0.5 is irrelevant); mean over ALL proposal points (panoptic_losses.py:156-201)."""
    dev = instance_labels.device
    pts, pid, n_prop = _proposal_rows(predicted_clusters, clusters_csr, dev)
    k, off, total = _gt_layout(instance_labels, batch)
    best, col = ious.max(1)
    # column -> instance id inside its batch element: col + 1 - (largest offset strictly below col + 1)
    idx1 = col + 1
    below = off[None, :] < idx1[:, None]
    base = torch.where(below, off[None, :], torch.full_like(off[None, :], -1)).max(1)[0]
    local_id = idx1 - base
    supervised = best > 0.5
    w = supervised[pid].float()
    target = torch.where(supervised[pid], (instance_labels.long()[pts] == local_id[pid]).float(), torch.full((pts.shape[0],), 0.5, device=dev))
    return torch.nn.functional.binary_cross_entropy(mask_scores_sigmoid.reshape(-1), target, weight=w)


def instance_iou_loss(ious, predicted_clusters, cluster_scores, instance_labels, batch, min_iou_threshold=0.25,
                      max_iou_threshold=0.75):
    """BCE of the proposal scores against a soft target: 0 below min_iou, 1 above max_iou, linear in between, from the
    best IoU of each proposal with any ground-truth instance (panoptic_losses.py:92-114)."""
    n_prop = predicted_clusters.n if hasattr(predicted_clusters, "n") else len(predicted_clusters)
    assert n_prop == cluster_scores.shape[0]
    best = ious.max(1)[0]
    ramp = (best - min_iou_threshold) / (max_iou_threshold - min_iou_threshold)
    # strict comparisons as in the reference: exactly min_iou -> ramp value 0, exactly max_iou -> ramp value 1
    target = torch.where(best > max_iou_threshold, torch.ones_like(best),
                         torch.where(best < min_iou_threshold, torch.zeros_like(best), ramp))
    return torch.nn.functional.binary_cross_entropy(cluster_scores, target)


def discriminative_loss_single(prediction, correct_label, feature_dim, delta_v=0.5, delta_d=1.5, param_var=1.0,
                               param_dist=1.0, param_reg=0.001):
    pred = prediction.reshape(-1, feature_dim)
    unique_labels, unique_id, counts = torch.unique(correct_label, return_inverse=True, return_counts=True)
    k = unique_labels.numel()
    zero = pred.new_zeros(())
    if k == 0:
        return zero, zero, zero, zero
    mu = scatter(pred, unique_id, dim=0, reduce="sum") / (counts.reshape(-1, 1) + 1e-8)
    distance = torch.norm(pred - gather(mu, unique_id), p=1, dim=1)
    distance = torch.square(torch.clip(distance - delta_v, min=0.0))
    l_var = scatter(distance, unique_id, dim=0, reduce="sum") / (counts + 1e-8)
    l_var = torch.sum(l_var) / float(k)
    if k > 1:
        diff = mu.unsqueeze(0) - mu.unsqueeze(1)  # all ordered pairs
        off_diag = ~torch.eye(k, dtype=torch.bool, device=pred.device)
        mu_norm = torch.norm(diff[off_diag], p=1, dim=1)
        l_dist = torch.mean(torch.square(torch.clip(2.0 * delta_d - mu_norm, min=0.0)))
    else:
        l_dist = zero
    l_reg = torch.mean(torch.norm(mu, p=1, dim=1))
    l_var, l_dist, l_reg = param_var * l_var, param_dist * l_dist, param_reg * l_reg
    return l_var + l_dist + l_reg, l_var, l_dist, l_reg


def discriminative_loss_per_sample(embedding_logits, instance_labels, batch, feature_dim):
    """the reference's formulation, literally: one discriminative_loss_single per batch element (reference
    models/panoptic/... discriminative loss loop); kept as the checker of the batched form below"""
    parts = []
    for s in torch.unique(batch):
        m = batch == s
        parts.append(discriminative_loss_single(gather(embedding_logits, m), instance_labels[m], feature_dim))
    loss, var, dist, reg = (torch.stack([p[i] for p in parts]) for i in range(4))
    return {"ins_loss": torch.mean(loss), "ins_var_loss": torch.mean(var), "ins_dist_loss": torch.mean(dist),
            "ins_reg_loss": torch.mean(reg)}


def discriminative_loss(embedding_logits, instance_labels, batch, feature_dim, delta_v=0.5, delta_d=1.5, param_var=1.0,
                        param_dist=1.0, param_reg=0.001):
    """Same four losses as discriminative_loss_per_sample, all batch elements at once: the clusters are the distinct
    (batch element, instance) keys, the pull term is a segment mean per cluster then per element, the push term runs over
    the ordered pairs of distinct clusters of the SAME element (block-diagonal pair list, not a C x C matrix over the
    whole batch), and the means over elements come last.  ~40 launches and 3 host reads instead of ~40 launches and 3
    host reads PER ELEMENT (2.8 ms of a 34 ms training step were spent in the loop with the GPU idle)."""
    pred = embedding_logits.reshape(-1, feature_dim)
    if pred.shape[0] == 0:
        return discriminative_loss_per_sample(embedding_logits, instance_labels, batch, feature_dim)
    dev = pred.device
    lab = instance_labels.long()
    sb, sid = torch.unique(batch.long(), return_inverse=True)
    S = sb.numel()
    lo = lab.min()
    span = lab.max() - lo + 1
    uk, cid, counts = torch.unique(sid * span + (lab - lo), return_inverse=True, return_counts=True)
    C = uk.numel()
    csample = torch.div(uk, span, rounding_mode="floor")       # element of every cluster (clusters are sorted by element)
    k_s = torch.zeros(S, dtype=torch.int64, device=dev).index_add_(0, csample, torch.ones_like(csample))  # clusters per element
    mu = scatter(pred, cid, dim=0, reduce="sum", dim_size=C, check=False) / (counts.reshape(-1, 1) + 1e-8)
    distance = torch.norm(pred - gather(mu, cid), p=1, dim=1)
    distance = torch.square(torch.clip(distance - delta_v, min=0.0))
    l_var_c = scatter(distance, cid, dim=0, reduce="sum", dim_size=C, check=False) / (counts + 1e-8)
    kf = k_s.to(pred.dtype)
    # (per-element sums through the library's segment sum: torch's index_add_ adds floats with atomics, in an order that
    # changes from run to run.  check=False throughout: cid / csample come out of torch.unique, so they are in range by
    # construction and the validation's host read -- one synchronisation per sum -- is skipped)
    l_var = scatter(l_var_c, csample, dim=0, reduce="sum", dim_size=S, check=False) / kf
    # ordered pairs (i, j), i != j, of clusters of the same element
    start = torch.cumsum(k_s, 0) - k_s
    reps = k_s[csample]
    i = torch.repeat_interleave(torch.arange(C, device=dev), reps)
    offs = torch.cumsum(reps, 0) - reps
    j = start[csample[i]] + (torch.arange(i.numel(), device=dev) - offs[i])
    keep = i != j
    i, j = i[keep], j[keep]
    if i.numel() > 0:
        mu_norm = torch.norm(mu[i] - mu[j], p=1, dim=1)
        h = torch.square(torch.clip(2.0 * delta_d - mu_norm, min=0.0))
        npairs = (kf * (kf - 1.0)).clamp_min(1.0)              # elements with one cluster: l_dist = 0
        l_dist = scatter(h, csample[i], dim=0, reduce="sum", dim_size=S, check=False) / npairs
    else:
        l_dist = torch.zeros(S, dtype=pred.dtype, device=dev)
    l_reg = scatter(torch.norm(mu, p=1, dim=1), csample, dim=0, reduce="sum", dim_size=S, check=False) / kf
    l_var, l_dist, l_reg = param_var * l_var, param_dist * l_dist, param_reg * l_reg
    return {"ins_loss": torch.mean(l_var + l_dist + l_reg), "ins_var_loss": torch.mean(l_var), "ins_dist_loss": torch.mean(l_dist),
            "ins_reg_loss": torch.mean(l_reg)}
