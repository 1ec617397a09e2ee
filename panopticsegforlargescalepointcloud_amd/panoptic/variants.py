"""The reference's two-head panoptic models on the same kernels as PointGroup3heads.

PointGroup      -- torch_points3d/models/panoptic/pointgroup.py:20-185 (semantic + offset heads; README settings II, III):
                   cluster_type 1 = region growing on shifted points, 2 = region growing on raw and on shifted points.
PointGroupEmbed -- torch_points3d/models/panoptic/pointgroupembed.py:33-500 (semantic + embedding heads; setting I):
                   cluster_type 7 = mean shift on the embeddings (:469-498, the published setting),
                   cluster_type 1 = HDBSCAN on raw coordinates united with HDBSCAN on the embeddings (:219-256),
                   cluster_type 14 = HDBSCAN on the embeddings alone (:683-710, starred in SURVEY.md section 2).
                   The remaining cluster types of that file (random feature subsets, :258-681, :712-783) are experiments
                   no published configuration selects; utils/hdbscan_cluster.cluster_loop is provided for them.
Sub-module names (hence state_dict keys) are the reference's: the absent head simply does not exist.
"""
import torch

from .. import ops
from ..utils import hdbscan_cluster
from .pointgroup3heads import PointGroup3heads


class PointGroup(PointGroup3heads):
    HEADS = ("Semantic", "Offset")

    def _cluster_fns(self):
        return {1: self._cluster, 2: self._cluster2}


class PointGroupEmbed(PointGroup3heads):
    HEADS = ("Semantic", "Embed")

    def _cluster_fns(self):
        return {1: self._cluster_hdbscan, 7: self._cluster7, 14: self._cluster14}

    def _thing_points(self, pred):
        label_mask = ~torch.isin(pred, self._stuff_classes.to(pred.device))
        return label_mask, torch.nonzero(label_mask).view(-1)

    def _cluster_hdbscan(self, pred, off, emb):
        mask, local_ind = self._thing_points(pred)
        batch = self.input.batch[mask]
        xyz = hdbscan_cluster.cluster_csr(self.raw_pos[mask], batch, local_ind, 3)
        embed = hdbscan_cluster.cluster_csr(emb[mask], batch, local_ind, 3)
        return ops.ClusterCSR.concat([xyz, embed]), self._types([(xyz, 0), (embed, 1)], pred.device)

    def _cluster14(self, pred, off, emb):
        """HDBSCAN on the embeddings of the thing points, one run per batch element with more than 3 of them; every
        proposal has cluster type 0 (pointgroupembed.py:683-710 -> utils/hdbscan_cluster.cluster_single :117-167)"""
        mask, local_ind = self._thing_points(pred)
        embed = hdbscan_cluster.cluster_csr(emb[mask], self.input.batch[mask], local_ind, 3)
        return embed, self._types([(embed, 0)], pred.device)

    def _cluster7(self, pred, off, emb):
        embed = self._embed_clusters(pred, emb)
        return embed, self._types([(embed, 0)], pred.device)
