"""The reference's two-head panoptic models on the same kernels as PointGroup3heads.

PointGroup      -- torch_points3d/models/panoptic/pointgroup.py:20-185 (semantic + offset heads; README settings II, III):
                   cluster_type 1 = region growing on shifted points, 2 = region growing on raw and on shifted points.
PointGroupEmbed -- torch_points3d/models/panoptic/pointgroupembed.py:33-500 (semantic + embedding heads; setting I):
                   cluster_type 1..16 (:121-152) as recipes over HDBSCAN / mean shift / region growing on the thing
                   points; 7 = mean shift on the embeddings (:469-498) is the published setting, 14 = HDBSCAN on the
                   embeddings alone (:683-710) is starred in SURVEY.md section 2.
Sub-module names (hence state_dict keys) are the reference's: the absent head simply does not exist.
"""
import torch

from .. import ops
from ..utils import hdbscan_cluster, meanshift_cluster
from .pointgroup3heads import PointGroup3heads


class PointGroup(PointGroup3heads):
    HEADS = ("Semantic", "Offset")

    def _cluster_fns(self):
        return {1: self._cluster, 2: self._cluster2}


class PointGroupEmbed(PointGroup3heads):
    """cluster_type 1..16 of pointgroupembed.py:121-152.  Every type is a union of proposal sets from a few primitives over
    the thing points (predicted class not in stuff_classes):
        H(x, t)          HDBSCAN per batch element (> 3 points), type t            hdbscan_cluster.cluster_single
        HL(x, lo, hi, n) n HDBSCAN runs on random feature subsets of size lo..hi,   hdbscan_cluster.cluster_loop
                         type = run index (elements with > 5 points)
        HF(x, n)         the same with subsets of 5 features                        hdbscan_cluster.cluster_loop_fixedD
        M(x, t)          mean shift (opt.bandwidth), type t                         meanshift_cluster.cluster_single
        ML(x, n)         n mean-shift runs on random 5-feature subsets              meanshift_cluster.cluster_loop
        R(t)             region growing on the raw positions, all points            torch_points_kernels.region_grow
    with xyz = raw positions, emb = embeddings, all = [xyz | emb].  The random subsets consume numpy's and torch's global
    CPU generators exactly as the reference does, so a seeded run selects the same features.
    Types 9, 10, 12 and 15 call ML, which raises TypeError in the reference (meanshift_cluster.py:57 calls
    meanshift_cluster without its bandwidth): they have no reference behaviour to compare with; here ML gets
    opt.bandwidth.  Type 12 lists the region-growing proposals FIRST but their types LAST (:638-641); kept as is."""
    HEADS = ("Semantic", "Embed")

    RECIPES = {
        1: [("H", "xyz", 0), ("H", "emb", 1)],                                   # :219-256
        2: [("HL", "all", 3, 5, 9), ("H", "emb", 9)],                            # :258-291
        3: [("HL", "all", 3, 5, 9), ("H", "xyz", 9)],                            # :294-327
        4: [("HL", "all", 3, 5, 8), ("H", "emb", 8), ("H", "xyz", 9)],           # :330-368
        5: [("HL", "all", 3, 5, 10)],                                            # :371-395
        6: [("HL", "emb", 2, 5, 6)],                                             # :398-421
        7: [("M", "emb", 0)],                                                    # :469-497 (the published setting I)
        8: [("R", 0), ("M", "emb", 1)],                                          # :500-545
        9: [("R", 0), ("ML", "emb", 10)],                                        # :424-466
        10: [("ML", "emb", 6)],                                                  # :548-571
        11: [("HF", "emb", 6)],                                                  # :574-597
        12: [("R", 6), ("ML", "emb", 6)],                                        # :600-644 (types listed in swapped order)
        13: [("HF", "emb", 6), ("H", "xyz", 6)],                                 # :647-681
        14: [("H", "emb", 0)],                                                   # :683-710
        15: [("ML", "emb", 6), ("H", "emb", 6)],                                 # :712-746
        16: [("M", "emb", 6), ("HL", "emb", 2, 5, 6)],                           # :749-783
    }

    def _cluster_fns(self):
        return {t: (lambda pred, off, emb, _t=t: self._cluster_recipe(_t, pred, emb)) for t in self.RECIPES}

    def _thing_points(self, pred):
        label_mask = ~torch.isin(pred, self._stuff_classes.to(pred.device))
        return label_mask, torch.nonzero(label_mask).view(-1)

    def _cluster_recipe(self, cluster_type, pred, emb):
        recipe = self.RECIPES[cluster_type]
        mask, local_ind = self._thing_points(pred)
        batch = self.input.batch[local_ind]
        feats = {"xyz": lambda: self.raw_pos[local_ind], "emb": lambda: emb[local_ind],
                 "all": lambda: torch.cat((self.raw_pos[local_ind], emb[local_ind]), 1)}
        parts = []
        for step in recipe:
            kind = step[0]
            if kind == "R":
                got = [(self._grow(self.raw_pos, pred, None), step[1])]
            elif kind == "H":
                got = [(hdbscan_cluster.cluster_csr(feats[step[1]](), batch, local_ind, 3), step[2])]
            elif kind == "HL":
                got = hdbscan_cluster.loop_csr(feats[step[1]](), batch, local_ind, hdbscan_cluster.loop_picks(*step[2:5]))
            elif kind == "HF":
                got = hdbscan_cluster.loop_csr(feats[step[1]](), batch, local_ind, [5] * step[2])
            elif kind == "M":
                got = [(meanshift_cluster.cluster_single_csr(feats[step[1]](), batch, local_ind, self.opt.bandwidth), step[2])]
            else:  # "ML"
                got = meanshift_cluster.loop_csr(feats[step[1]](), batch, local_ind, step[2], self.opt.bandwidth)
            parts.append(got)
        proposals, typed = self._order(cluster_type, parts)
        csr = ops.ClusterCSR.concat(proposals) if len(proposals) > 1 else proposals[0]
        return csr, self._types(typed, pred.device)

    @staticmethod
    def _order(cluster_type, parts):
        """parts: per recipe step a list of (proposal set, type).  Returns (the sets in output order, the (set, type) pairs
        in the order their type codes are listed) -- identical orders except for type 12, whose region-growing proposals
        come first while their type codes come last (:638-641)."""
        flat = [cp for got in parts for cp in got]
        typed = flat if cluster_type != 12 else [cp for got in parts[1:] for cp in got] + parts[0]
        return [c for c, _ in flat], typed
