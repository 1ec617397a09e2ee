"""PanopticResults / PanopticLabels with the reference's field names and get_instances semantics
(torch_points3d/models/panoptic/structure_3heads.py:6-71): proposal x proposal IoU, greedy NMS at `nms_threshold`,
then the size and score filters.  The dense [nProp, N] mask matmul of the reference (:40-60) and the host NMS loop are
replaced by device kernels (csrc/pp_nms.hip: incidence pass, overlapping pairs, greedy NMS, ranks)."""
from typing import List, NamedTuple, Optional

import numpy as np
import torch

from .. import ops


def non_max_suppression(ious, scores, threshold):
    """Greedy NMS over a dense IoU matrix (host form of structure_3heads.py:6-16, kept for callers that hold the dense
    matrix; the model path uses the device kernels behind ops.nms_paint).  Visits proposals by descending score
    and drops everything a kept proposal overlaps by > threshold.  Equal scores: descending index (stable sort reversed --
    numpy's `argsort()[::-1]` on up to 16 elements; beyond that the reference's introsort leaves ties undefined)."""
    order = np.argsort(scores, kind="stable")[::-1]
    alive = np.ones(len(order), dtype=bool)
    kept = []
    for i in order:
        if alive[i]:
            kept.append(i)
            alive &= ~(ious[i] > threshold)
            alive[i] = False
    return kept


def masked_csr(csr, mask_scores):
    """proposals restricted to the points whose mask logit exceeds -0.5 (structure_3heads.py:36-54, the mask-supervised
    scorer): same proposals, fewer points each.  mask_scores: one logit per proposal ROW ([R] or [R, 1])."""
    if mask_scores is None:
        return csr
    keep = mask_scores.reshape(-1) > -0.5
    sizes = csr.sizes().long()
    pid = torch.repeat_interleave(torch.arange(csr.n, device=keep.device), sizes, output_size=keep.shape[0])
    kept = torch.zeros(csr.n, dtype=torch.int64, device=keep.device).index_add_(0, pid[keep], torch.ones(int(keep.sum()), dtype=torch.int64, device=keep.device))
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=keep.device), torch.cumsum(kept, 0)]).to(torch.int32)
    return ops.ClusterCSR(offsets, csr.points[keep], csr.n)


class PanopticResults(NamedTuple):
    semantic_logits: torch.Tensor
    offset_logits: torch.Tensor
    embed_logits: torch.Tensor
    cluster_scores: Optional[torch.Tensor]  # one float per cluster
    mask_scores: Optional[torch.Tensor]
    clusters: Optional[List[torch.Tensor]]  # point indices of each cluster
    cluster_type: Optional[torch.Tensor]  # 0 -> original pos, 1 -> vote / embedding, 2 -> embedding (type 6)
    clusters_csr: Optional[ops.ClusterCSR] = None  # device-resident form of `clusters` (build-owned extra field)

    def _csr(self):
        if self.clusters_csr is not None:
            return self.clusters_csr
        return ops.ClusterCSR.from_list(self.clusters, self.semantic_logits.device)

    def get_instances(self, nms_threshold=0.3, min_cluster_points=100, min_score=0.5):
        """Returns (indices of clusters that pass NMS + size + score tests, their point lists), best score first --
        structure_3heads.py:28-71 with the dense [nProp, N] mask product replaced by the device kernels of
        csrc/pp_nms.hip (the whole batch is one NMS group here, exactly like the reference's joint IoU matrix)."""
        if not self.clusters and (self.clusters_csr is None or self.clusters_csr.n == 0):
            return [], []
        if self.cluster_scores is None:
            return None, self.clusters if self.clusters is not None else self._csr().to_list()
        csr = masked_csr(self._csr(), self.mask_scores)
        n_points = self.semantic_logits.shape[0]
        _, _, rank, pairs = ops.nms_paint(csr, n_points, None, 1, self.cluster_scores, nms_threshold, min_cluster_points,
                                          min_score)
        rank = rank.cpu().numpy()
        pairs.check()
        kept = np.nonzero(rank >= 0)[0]
        valid_pick_ids = [int(i) for i in kept[np.argsort(-rank[kept], kind="stable")]]  # pick order = descending score
        clusters = self.clusters if (self.clusters is not None and self.mask_scores is None) else csr.to_list()
        return valid_pick_ids, [clusters[i] for i in valid_pick_ids]


class PanopticLabels(NamedTuple):
    center_label: torch.Tensor
    y: torch.Tensor
    num_instances: torch.Tensor
    instance_labels: torch.Tensor
    instance_mask: torch.Tensor
    vote_label: torch.Tensor
