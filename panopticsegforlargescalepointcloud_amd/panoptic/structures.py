"""PanopticResults / PanopticLabels with the reference's field names and get_instances semantics
(torch_points3d/models/panoptic/structure_3heads.py:6-71): proposal x proposal IoU, greedy NMS at `nms_threshold`,
then the size and score filters.  The dense [nProp, N] mask matmul of the reference (:40-60) is replaced by the
point->proposal incidence kernel (pp_proposal_intersections); the greedy pick itself stays on the host as in the
reference (non_max_suppression works on numpy there too)."""
from typing import List, NamedTuple, Optional

import numpy as np
import torch

from .. import ops


def non_max_suppression(ious, scores, threshold):
    """Greedy NMS over a dense IoU matrix (host form of structure_3heads.py:6-16, kept for callers that hold the dense
    matrix; the model path uses the device kernels behind ops.nms_paint).  Visits proposals by descending score
    (`argsort()[::-1]`, i.e. the reference's tie order) and drops everything a kept proposal overlaps by > threshold."""
    order = np.argsort(scores)[::-1]
    alive = np.ones(len(order), dtype=bool)
    kept = []
    for i in order:
        if alive[i]:
            kept.append(i)
            alive &= ~(ious[i] > threshold)
            alive[i] = False
    return kept


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
        """Returns (indices of clusters that pass NMS + size + score tests, their point lists)."""
        if not self.clusters and (self.clusters_csr is None or self.clusters_csr.n == 0):
            return [], []
        if self.cluster_scores is None:
            return None, self.clusters if self.clusters is not None else self._csr().to_list()
        csr = self._csr()
        n_points = self.semantic_logits.shape[0]
        inter = ops.proposal_intersections(csr, n_points).cpu().numpy().astype(np.float32)
        sizes = np.diag(inter).copy()
        cross_ious = inter / (sizes[:, None] + sizes[None, :] - inter)
        pick_idxs = non_max_suppression(cross_ious, self.cluster_scores.detach().cpu().numpy(), nms_threshold)
        scores = self.cluster_scores.detach().cpu().numpy()
        clusters = self.clusters if self.clusters is not None else csr.to_list()
        valid_pick_ids, valid_clusters = [], []
        for i in pick_idxs:
            if sizes[i] > min_cluster_points and scores[i] > min_score:
                valid_pick_ids.append(i)
                valid_clusters.append(clusters[i])
        return valid_pick_ids, valid_clusters


class PanopticLabels(NamedTuple):
    center_label: torch.Tensor
    y: torch.Tensor
    num_instances: torch.Tensor
    instance_labels: torch.Tensor
    instance_mask: torch.Tensor
    vote_label: torch.Tensor
