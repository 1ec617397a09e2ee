"""PointGroup3heads -- the reference's 3-head panoptic model (semantic + offset + embedding heads on a sparse
U-Net, region growing / mean shift proposals, ScorerUnet), on the MI355X kernels.

Mirrors torch_points3d/models/panoptic/PointGroup3heads.py: constructor arguments and sub-module names
(=> state_dict keys of SURVEY.md App. A), set_input :95-99, forward :101-157, _cluster/_cluster2/_cluster3/_cluster4/
_cluster5/_cluster6 :163-391, _compute_score :393-454, _compute_loss :552-634, backward :636-639.

What is different by design (MI355X-first, results unchanged): everything stays device-resident -- proposals travel
as CSR (ops.ClusterCSR) instead of Python lists until the caller asks for lists; mean shift runs for all cylinders of
the batch at once on the GPU instead of one sklearn process per cylinder; the scorer batch is assembled with one
gather instead of a Python loop over proposals; eval-mode layers run as fused launches.
"""
import os
import threading
from collections import OrderedDict

import torch
from torch import nn

from .. import MinkowskiEngine as ME
from .. import ops
from ..applications import Data, Minkowski
from ..modules import MLP, Linear, Seq, fused_head, head_spec
from ..torch_points_kernels import region_grow_csr
from ..torch_scatter import gather, scatter
from ..utils import meanshift_cluster
from .losses import discriminative_loss, instance_iou_loss, instance_ious, mask_loss, offset_loss, semantic_nll
from .structures import PanopticLabels, PanopticResults

IGNORE_LABEL = -1  # torch_points3d/datasets/segmentation/__init__.py
MAX_SCORER_BATCH = 60000  # proposals per ScorerUnet launch (batch index must fit the 16-bit key field)
FUSE_HEADS = os.environ.get("PP_FUSE_HEADS", "1") != "0"  # all heads in one pass over the un-permuted features
OVERLAP_CLUSTERING = os.environ.get("PP_CLUSTER_OVERLAP", "1") != "0"  # mean shift on a side stream next to region growing


# duplicate proposals + the scorer's input batch from csrc/pp_proposals.hip (PP_DEDUPE_FUSED=0: the tensor-library form, A/B runs)
DEDUPE_FUSED = os.environ.get("PP_DEDUPE_FUSED", "1") != "0"
DEDUPE_HASH = os.environ.get("PP_DEDUPE_HASH", "1") != "0"


def _duplicate_representatives_from_pairs(csr, n_points):
    """the first form: two proposals are identical when their intersection equals both sizes (ops.proposal_pairs)"""
    pairs = ops.proposal_pairs(csr, n_points)  # device-side; reused by the NMS of this batch
    sz = csr.sizes()
    n_prop = csr.n
    valid = torch.arange(pairs.capacity, device=sz.device) < pairs.n_pairs
    a = torch.where(valid, pairs.a.long(), 0)
    b = torch.where(valid, pairs.b.long(), 0)
    inter = pairs.inter.long()
    dup = valid & (inter == sz[a]) & (inter == sz[b])
    # non-duplicates go to private slots behind the proposals (one shared dump slot would serialise ~10^5 atomics)
    slot = torch.arange(pairs.capacity, device=sz.device)
    rep = torch.arange(n_prop + pairs.capacity, device=sz.device)
    rep.scatter_reduce_(0, torch.where(dup, b, n_prop + slot), torch.where(dup, a, n_prop + slot), "amin", include_self=True)
    return rep[:n_prop]


def _duplicate_representatives(csr):
    """rep int64 [P]: the smallest index of a proposal with exactly the same point list (rep[p] == p for the first of its kind).
    Device only, no host synchronisation.  Candidates = equal (size, hash1, hash2) with two independent 64-bit sum hashes of the
    point ids (order-free, exact integer arithmetic); each candidate is verified entry by entry against its representative."""
    dev = csr.points.device
    n = csr.n
    sz = csr.sizes().long()
    pts = csr.points.long()
    total = int(pts.numel())
    ids = torch.arange(n, device=dev)
    owner = torch.repeat_interleave(ids, sz, output_size=total)
    m1 = (pts + 0x165667B19E3779F9) * -0x61C8864680B583EB          # (int64 arithmetic wraps: that is the hash)
    m1 = m1 ^ (m1 >> 29)
    m2 = (pts ^ 0x27D4EB2F165667C5) * -0x3A8F057B4E4F7C2B
    m2 = m2 ^ (m2 >> 31)
    offs = csr.offsets.long()
    zero = torch.zeros(1, dtype=torch.int64, device=dev)

    def seg_sum(v):  # per-proposal sums as differences of one prefix sum (wrapping int64 arithmetic: exact); an index_add_ would
        cs = torch.cat([zero, torch.cumsum(v, 0)])  # serialise ~10^3 atomics per address
        return cs[offs[1:]] - cs[offs[:-1]]
    h1, h2 = seg_sum(m1), seg_sum(m2)
    key = h1 ^ (h2 * -0x4B47B4E4F7C2B1D5) ^ (sz * -0x61C8864680B583EB)
    skey, order = torch.sort(key, stable=True)                       # equal keys keep ascending proposal index
    same_as_prev = torch.zeros(n, dtype=torch.bool, device=dev)
    same_as_prev[1:] = (skey[1:] == skey[:-1]) & (sz[order][1:] == sz[order][:-1]) & (h1[order][1:] == h1[order][:-1]) \
        & (h2[order][1:] == h2[order][:-1])
    start_pos = torch.cummax(torch.where(same_as_prev, torch.zeros_like(ids), ids), 0)[0]   # position of the group's first member
    rep = torch.empty(n, dtype=torch.int64, device=dev)
    rep[order] = order[start_pos]
    # exact verification: entry j of p against entry j of rep[p]
    pos_in = torch.arange(total, device=dev) - offs[owner]
    other = pts[(offs[rep[owner]] + pos_in).clamp_(max=total - 1)]
    mismatch = seg_sum((other != pts).long())
    return torch.where(mismatch > 0, ids, rep)


class PointGroup3heads(nn.Module):
    HEADS = ("Semantic", "Offset", "Embed")  # the 2-head classes of settings I-III narrow this (panoptic/variants.py)
    __REQUIRED_DATA__ = ["pos"]
    __REQUIRED_LABELS__ = list(PanopticLabels._fields)

    def __init__(self, option, model_type, dataset, modules):
        super().__init__()
        self.opt = option
        backbone_options = option.get("backbone", {"architecture": "unet"})
        self.Backbone = Minkowski(backbone_options.get("architecture", "unet"), input_nc=dataset.feature_dimension,
                                  num_layers=4, config=backbone_options.get("config", {}))
        self._scorer_type = option.get("scorer_type", None)
        self._voxelizer = None
        nc = self.Backbone.output_nc
        self.ScorerUnet = Minkowski("unet", input_nc=nc, num_layers=4, config=option.scorer_unet)
        self.ScorerEncoder = Minkowski("encoder", input_nc=nc, num_layers=4, config=option.scorer_encoder)
        self.ScorerMLP = MLP([nc, nc, self.ScorerUnet.output_nc])
        self.ScorerHead = Seq().append(nn.Linear(self.ScorerUnet.output_nc, 1)).append(nn.Sigmoid())
        self.mask_supervise = option.get("mask_supervise", False)
        if self.mask_supervise:
            self.MaskScore = (Seq().append(nn.Linear(self.ScorerUnet.output_nc, self.ScorerUnet.output_nc))
                              .append(nn.ReLU()).append(nn.Linear(self.ScorerUnet.output_nc, 1)))
        self.use_score_net = option.get("use_score_net", True)
        self.use_mask_filter_score_feature = option.get("use_mask_filter_score_feature", False)
        self.use_mask_filter_score_feature_start_epoch = option.get("use_mask_filter_score_feature_start_epoch", 200)
        self.mask_filter_score_feature_thre = option.get("mask_filter_score_feature_thre", 0.5)
        self.dedupe_proposals = True  # eval-only optimisation with identical results, see _compute_score
        self._side_streams = {}        # device -> HIP stream of the overlapped mean shift (see _embed_clusters_async)
        self.cal_iou_based_on_mask = option.get("cal_iou_based_on_mask", False)
        self.cal_iou_based_on_mask_start_epoch = option.get("cal_iou_based_on_mask_start_epoch", 200)

        if "Offset" in self.HEADS:
            self.Offset = Seq().append(MLP([nc, nc], bias=False))
            self.Offset.append(Linear(nc, 3))
        if "Embed" in self.HEADS:
            self.Embed = Seq().append(MLP([nc, nc], bias=False))
            self.Embed.append(Linear(nc, option.get("embed_dim", 5)))
        self.Semantic = (Seq().append(MLP([nc, nc], bias=False)).append(Linear(nc, dataset.num_classes))
                         .append(nn.LogSoftmax(dim=-1)))
        self.num_classes = dataset.num_classes
        self.loss_names = ["loss", "offset_norm_loss", "offset_dir_loss", "ins_loss", "ins_var_loss", "ins_dist_loss",
                           "ins_reg_loss", "semantic_loss", "score_loss", "mask_loss"]
        stuff_classes = dataset.stuff_classes
        if isinstance(stuff_classes, (list, tuple)):
            stuff_classes = torch.Tensor(stuff_classes).long()
        self._stuff_classes = torch.cat([torch.tensor([IGNORE_LABEL]), stuff_classes.long()])
        self.output = None

    # ------------------------------------------------------------------ BaseModel contract (models/base_model.py:136-297)
    @property
    def device(self):
        return next(self.parameters()).device

    def get_opt_mergeTh(self):
        return self.opt.block_merge_th if self.opt.get("block_merge_th") else 0.01

    def set_input(self, data, device):
        self.input = data.to(device)  # everything device-resident (the reference leaves `input` on the host)
        self.raw_pos = self.input.pos
        if all(hasattr(self.input, l) for l in self.__REQUIRED_LABELS__):
            self.labels = PanopticLabels(**{l: getattr(self.input, l) for l in self.__REQUIRED_LABELS__})
        else:
            self.labels = None

    def get_output(self):
        return self.output

    def get_labels(self):
        return self.labels

    def get_current_losses(self):
        out = OrderedDict()
        for name in self.loss_names:
            if hasattr(self, name):
                try:
                    out[name] = float(getattr(self, name))
                except Exception:
                    out[name] = None
        return out

    # ------------------------------------------------------------------ forward
    def backbone_and_heads(self, data=None):
        """Sparse U-Net + the three heads.  Returns (features [N,16], semantic log-probs, offsets, embeddings,
        predicted labels [N] int64).  data: the batch to run instead of the one `set_input` stored (scene.TileRunner runs the
        NEXT batch's backbone from a thread of its own while `self.input` still belongs to the batch being grouped)."""
        has_off, has_emb = "Offset" in self.HEADS, "Embed" in self.HEADS
        inp = self.input if data is None else data
        if not self.training and not torch.is_grad_enabled() and self.Backbone.output_nc == 16 and FUSE_HEADS:
            # inference: the backbone's features stay in the coordinate manager's row order and ALL heads run as one pass
            # that reads row inv_perm[i] for point i -- no un-permuting gather of the features, one read of them instead of
            # three; `feats` is handed on as "rows inv_perm of the internal matrix" (the scorer composes it with its own
            # proposal gather)
            out = self.Backbone(inp, internal_order=True)
            cm = self.Backbone.input.coordinate_manager
            specs = [head_spec(self.Semantic, True, True)] + ([head_spec(self.Offset)] if has_off else []) \
                + ([head_spec(self.Embed)] if has_emb else [])
            res = ops.heads(out.x, specs, index=cm.inv_perm)
            (sem, pred) = res[0]
            off = res[1][0] if has_off else None
            emb = res[1 + int(has_off)][0] if has_emb else None
            feats = out.x if cm.inv_perm is None else ME.GatheredRows(out.x, cm.inv_perm)
            return feats, sem, off, emb, pred
        feats = self.Backbone(inp).x
        if not self.training and not torch.is_grad_enabled():
            sem, pred = fused_head(self.Semantic, feats, log_softmax=True, want_argmax=True)
            off = fused_head(self.Offset, feats) if has_off else None
            emb = fused_head(self.Embed, feats) if has_emb else None
        else:
            sem = self.Semantic(feats)
            off = self.Offset(feats) if has_off else None
            emb = self.Embed(feats) if has_emb else None
            pred = torch.max(sem, 1)[1]
        return feats, sem, off, emb, pred

    def group_and_score(self, epoch, feats, sem, off, emb, pred=None, timer=None, t0=0.0):
        """Instance grouping (+ scoring) on given head outputs; returns a PanopticResults.
        timer(name, t0) -> t0: optional stage stopwatch used by bench.py (synchronises; off by default)."""
        self._timer, self._t0 = timer, t0
        if pred is None:
            pred = torch.max(sem, 1)[1]
        cluster_scores = mask_scores = csr = cluster_type = None
        ct = self.opt.cluster_type
        fns = self._cluster_fns()
        run = (epoch > self.opt.prepare_epoch) if self.use_score_net else True
        if run:
            if ct not in fns:
                raise NotImplementedError("%s: cluster_type %s (available: %s)" % (type(self).__name__, ct, sorted(fns)))
            with torch.no_grad():
                csr, cluster_type = fns[ct](pred, None if off is None else off.detach(),
                                            None if emb is None else emb.detach())
            if self.use_score_net and csr.n:
                # (the scorer's launches depend on this rank's proposals -- none at all, or several chunks -- so its BatchNorms never
                # take part in SyncBN's collectives: per-replica statistics over the rank's own proposals, ops.sync_bn_suspended)
                with ops.sync_bn_suspended():
                    cluster_scores, mask_scores = self._compute_score(epoch, csr, feats, sem)
                self._lap("scorer")
        return PanopticResults(semantic_logits=sem, offset_logits=off, embed_logits=emb, clusters=None,
                               cluster_scores=cluster_scores, mask_scores=mask_scores, cluster_type=cluster_type,
                               clusters_csr=csr)

    def forward(self, epoch=-1, **kwargs):
        feats, sem, off, emb, pred = self.backbone_and_heads()
        res = self.group_and_score(epoch, feats, sem, off, emb, pred)
        if res.clusters_csr is not None:  # materialise the reference's List[LongTensor] view
            res = res._replace(clusters=res.clusters_csr.to_list())
        self.output = res
        return res

    # ------------------------------------------------------------------ proposal generators
    def _cluster_fns(self):
        return {1: self._cluster, 2: self._cluster2, 3: self._cluster3, 4: self._cluster4, 5: self._cluster5, 6: self._cluster6}

    def _grow(self, pos, pred, nsample):
        kw = {} if nsample is None else {"nsample": nsample}  # reference leaves the default (16) for raw coordinates
        csr, _ = region_grow_csr(pos, pred, self.input.batch, ignore_labels=self._stuff_classes,
                                 radius=self.opt.cluster_radius_search, min_cluster_size=10,
                                 num_classes=self.num_classes, **kw)
        return csr

    def _embed_clusters(self, pred, emb):
        label_mask = ops.not_ignored(pred, self._stuff_classes, self.num_classes)
        local_ind = ops.select_indices(label_mask)  # (torch.nonzero on the library's own scan: no rocPRIM launch in the step)
        # (rows by index: a boolean mask costs another compaction pass and host synchronisation per use)
        return meanshift_cluster.cluster_single_csr(emb[local_ind], self.input.batch[local_ind], local_ind,
                                                    self.opt.bandwidth)

    @staticmethod
    def _types(parts, dev):
        return torch.cat([torch.full((c.n,), t, dtype=torch.uint8, device=dev) for c, t in parts])

    def _cluster(self, pred, off, emb):
        votes = self._grow(self.raw_pos + off, pred, 200)
        return votes, self._types([(votes, 0)], pred.device)

    def _cluster2(self, pred, off, emb):
        pos = self._grow(self.raw_pos, pred, None)
        votes = self._grow(self.raw_pos + off, pred, 200)
        # (the reference marks the votes as type 1 only when there are position clusters: PointGroup3heads.py:208-210)
        return ops.ClusterCSR.concat([pos, votes]), self._types([(pos, 0), (votes, 1 if pos.n else 0)], pred.device)

    def _cluster3(self, pred, off, emb):
        """mean shift on the embeddings of the thing points alone (reference :213-243; `off` is unused: the reference's
        signature is (semantic_logits, embed_logits))"""
        embed = self._embed_clusters(pred, emb)
        return embed, self._types([(embed, 0)], pred.device)

    def _cluster4(self, pred, off, emb):
        """region growing on the raw positions (library-default nsample) + mean shift on the embeddings (reference :246-289)"""
        pending = self._embed_clusters_async(pred, emb)
        pos = self._grow(self.raw_pos, pred, None)
        embed = pending() if pending is not None else self._embed_clusters(pred, emb)
        return ops.ClusterCSR.concat([pos, embed]), self._types([(pos, 0), (embed, 1)], pred.device)

    def _lap(self, name):
        if getattr(self, "_timer", None) is not None:
            self._t0 = self._timer(name, self._t0)

    def _embed_clusters_async(self, pred, emb):
        """Mean shift on the embeddings does not depend on region growing: in inference it runs on a side stream from a
        worker thread (both stages are chains of small launches separated by host synchronisations on data-dependent
        sizes, so one hides in the other's gaps).  Returns a callable that joins and yields the ClusterCSR."""
        if (not OVERLAP_CLUSTERING or torch.is_grad_enabled() or getattr(self, "_timer", None) is not None
                or not pred.is_cuda):
            return None
        main = torch.cuda.current_stream(pred.device)
        side = self._side_streams.get(pred.device)
        if side is None:
            side = self._side_streams[pred.device] = torch.cuda.Stream(device=pred.device)
        side.wait_stream(main)
        box = {}

        def work():
            try:
                with torch.cuda.device(pred.device), torch.cuda.stream(side), torch.no_grad():
                    box["csr"] = self._embed_clusters(pred, emb)
            except BaseException as e:  # re-raised by the caller's join
                box["err"] = e

        th = threading.Thread(target=work, name="pp-meanshift")
        th.start()

        def join():
            th.join()
            main.wait_stream(side)
            if "err" in box:
                raise box["err"]
            return box["csr"]
        return join

    def _cluster5(self, pred, off, emb):
        pending = self._embed_clusters_async(pred, emb)
        votes = self._grow(self.raw_pos + off, pred, 200)
        self._lap("region_grow")
        embed = pending() if pending is not None else self._embed_clusters(pred, emb)
        self._lap("meanshift")
        return ops.ClusterCSR.concat([votes, embed]), self._types([(votes, 0), (embed, 1)], pred.device)

    def _cluster6(self, pred, off, emb):
        pending = self._embed_clusters_async(pred, emb)
        pos = self._grow(self.raw_pos, pred, None)
        votes = self._grow(self.raw_pos + off, pred, 200)
        embed = pending() if pending is not None else self._embed_clusters(pred, emb)
        return (ops.ClusterCSR.concat([pos, votes, embed]),
                self._types([(pos, 0), (votes, 1), (embed, 2)], pred.device))

    # ------------------------------------------------------------------ scorer
    def _compute_score(self, epoch, csr, backbone_features, semantic_logits):
        if not self._scorer_type:
            with torch.no_grad():
                sizes = csr.sizes()
                b = torch.repeat_interleave(torch.arange(csr.n, device=sizes.device), sizes)
                mean_sem = scatter(semantic_logits[csr.points], b, dim=0, reduce="mean", dim_size=csr.n)
                return torch.max(torch.exp(mean_sem), 1)[0], None
        if self._scorer_type not in ("unet", "MLP", "encoder"):
            raise NotImplementedError("scorer_type %s (the reference knows 'unet', 'MLP' and 'encoder')" % self._scorer_type)
        if self.dedupe_proposals and not torch.is_grad_enabled() and csr.n > 1 and not self.mask_supervise:
            # Region growing and mean shift often return the SAME point set for a well-separated instance; identical
            # proposals get identical ScorerUnet inputs, hence identical scores: score one representative per set.
            # Round 4: duplicates are found through (size, two 64-bit sum hashes of the point ids) -- three passes over the
            # CSR entries and a sort of the ~10^3 proposals -- and every candidate is then verified entry by entry against its
            # representative, so the result is exact (a mismatch, i.e. a hash collision or differently ordered lists, simply
            # keeps the proposal).  The overlap-pair pass the first form was built on (0.9 ms, needed by NMS anyway) no longer
            # sits in front of the scorer: it runs on a side stream next to the scorer's convolutions.
            if DEDUPE_FUSED and csr.n <= MAX_SCORER_BATCH and self._scorer_type == "unet":
                # the kept lists, their batch index and the scorer's coordinate rows from csrc/pp_proposals.hip: seven launches and
                # one host read where the tensor-library form below needs ~130 launches (5 ms of the bench step in which the
                # GPU waits for the host to issue them)
                coords = self.input.coords if self.input.coords.dtype == torch.int32 else self.input.coords.int()
                uniq = ops.proposals_unique(csr, backbone_features.shape[0], coords=coords.contiguous())
                self._pairs_async(csr, backbone_features.shape[0])
                scores_u, _ = self._score_unique(uniq.csr, backbone_features, prepared=uniq)
                return scores_u[uniq.pos_of], None
            if DEDUPE_HASH:
                rep = _duplicate_representatives(csr)
                self._pairs_async(csr, backbone_features.shape[0])
            else:  # round-3 form (A/B runs): duplicates from the overlap pairs, on the scorer's critical path
                rep = _duplicate_representatives_from_pairs(csr, backbone_features.shape[0])
            uniq_ids = torch.nonzero(rep == torch.arange(csr.n, device=rep.device)).view(-1)
            if uniq_ids.numel() < csr.n:
                pos_of = torch.empty(csr.n, dtype=torch.int64, device=rep.device)
                pos_of[uniq_ids] = torch.arange(uniq_ids.numel(), device=rep.device)
                scores_u, _ = self._score_unique(csr.select(uniq_ids), backbone_features)
                return scores_u[pos_of[rep]], None
        return self._score_unique(csr, backbone_features, epoch)

    def _pairs_async(self, csr, n_points):
        """ops.proposal_pairs(csr) on the clustering side stream: NMS (scene.instance_labels_per_tile -> ops.nms_paint) finds the
        table cached on the csr; its consumer stream waits for the event recorded here."""
        if getattr(csr, "_pairs", None) is not None or not csr.points.is_cuda or not OVERLAP_CLUSTERING:
            return
        dev = csr.points.device
        main = torch.cuda.current_stream(dev)
        side = self._side_streams.get(dev)
        if side is None:
            side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pairs = ops.proposal_pairs(csr, n_points)
            ev = torch.cuda.Event()
            ev.record(side)
        for t in (pairs.a, pairs.b, pairs.inter, pairs.n_pairs, pairs.prop_of_entry, pairs.info):
            t.record_stream(main)  # allocated on the side stream, read by the NMS kernels of the main stream
        pairs.ready = ev

    def _score_unique(self, csr, backbone_features, epoch=-1, prepared=None):
        """prepared (ops.UniqueProposals of this very csr, one chunk): batch index and coordinate rows are already there"""
        sizes = csr.sizes()
        offsets = csr.offsets.long()
        scores, masks = [], []
        for lo in range(0, csr.n, MAX_SCORER_BATCH):
            hi = min(lo + MAX_SCORER_BATCH, csr.n)
            if lo == 0 and hi == csr.n:  # one chunk: all entries, no host read of the offsets
                p0, p1 = 0, int(csr.points.numel())
            else:
                p0, p1 = (int(v) for v in offsets[[lo, hi]].tolist())
            pts = csr.points[p0:p1]
            one_chunk = prepared is not None and lo == 0 and hi == csr.n
            b = prepared.batch if one_chunk else \
                torch.repeat_interleave(torch.arange(hi - lo, device=pts.device), sizes[lo:hi], output_size=p1 - p0)
            # one gather for "rows of the proposals" + "internal row order", none for the way back (the max is order-free)
            if self._scorer_type == "MLP":
                # per-point MLP on the proposals' backbone rows, then the per-proposal maximum (reference :419-423)
                rows = backbone_features[pts] if not isinstance(backbone_features, ME.GatheredRows) \
                    else ME.GatheredRows(backbone_features, pts).materialise()
                cluster_feats = scatter(self.ScorerMLP(rows), b, dim=0, reduce="max", dim_size=hi - lo)
            elif self._scorer_type == "encoder":
                # sparse encoder with a global max-pool head: one feature row per proposal, in batch order (:424-426)
                batch_cluster = Data(x=ME.GatheredRows(backbone_features, pts), coords=self.input.coords[pts], batch=b, pos=None)
                cluster_feats = self.ScorerEncoder(batch_cluster).x
                if cluster_feats.shape[0] != hi - lo:
                    raise RuntimeError("ScorerEncoder returned %d rows for %d proposals" % (cluster_feats.shape[0], hi - lo))
            elif self.mask_supervise:
                # mask-supervised scorer (reference :427-436): one mask logit per proposal ROW, in the order the proposals
                # were concatenated (losses and trackers slice it by proposal) -- so the U-Net output is un-permuted here
                batch_cluster = Data(x=ME.GatheredRows(backbone_features, pts), coords=self.input.coords[pts], batch=b, pos=None)
                out = self.ScorerUnet(batch_cluster)
                mask = self.MaskScore(out.x)
                masks.append(mask)
                x = out.x
                if self.use_mask_filter_score_feature and epoch > self.use_mask_filter_score_feature_start_epoch:
                    x = x * (torch.sigmoid(mask) >= self.mask_filter_score_feature_thre).to(x.dtype)
                cluster_feats = scatter(x, b, dim=0, reduce="max", dim_size=hi - lo)
            elif one_chunk and prepared.coords4 is not None:
                c4 = prepared.coords4
                batch_cluster = Data(x=ME.GatheredRows(backbone_features, pts), coords=c4[:, 1:], coords4=c4, batch=b, pos=None)
                out = self.ScorerUnet(batch_cluster, internal_order=True)
                # (batch ids written by pp_proposals_emit: no range check, i.e. no host synchronisation between the scorer's last
                # convolution and the head / NMS launches -- the step's one read at its end is the next time the host waits)
                cluster_feats = scatter(out.x, out.batch.long(), dim=0, reduce="max", dim_size=hi - lo, check=False)
            else:
                batch_cluster = Data(x=ME.GatheredRows(backbone_features, pts), coords=self.input.coords[pts], batch=b, pos=None)
                out = self.ScorerUnet(batch_cluster, internal_order=True)
                cluster_feats = scatter(out.x, out.batch.long(), dim=0, reduce="max", dim_size=hi - lo)
            # Linear(16, 1) + Sigmoid written as a reduction: a [P,16]x[16,1] GEMM goes through hipBLASLt, whose
            # dispatch costs milliseconds of host time per call for 0.1 ms of work
            lin = self.ScorerHead[0]
            scores.append(torch.sigmoid((cluster_feats * lin.weight[0]).sum(1) + lin.bias[0]))
        mask_scores = None if not masks else (masks[0] if len(masks) == 1 else torch.cat(masks))
        return (scores[0] if len(scores) == 1 else torch.cat(scores)), mask_scores

    # ------------------------------------------------------------------ losses / backward
    def _compute_loss(self, epoch):
        out, inp = self.output, self.input
        self.semantic_loss = semantic_nll(out.semantic_logits, self.labels.y.to(torch.int64), IGNORE_LABEL)
        self.loss = self.opt.loss_weights.semantic * self.semantic_loss
        mask = inp.instance_mask
        rows = torch.nonzero(mask).view(-1)  # ONE compaction (a host read each) for the five row selections below
        if out.offset_logits is not None:
            for name, loss in offset_loss(gather(out.offset_logits, rows), inp.vote_label[rows], torch.sum(mask)).items():
                setattr(self, name, loss)
                self.loss = self.loss + self.opt.loss_weights[name] * loss
        if out.embed_logits is not None:
            for name, loss in discriminative_loss(gather(out.embed_logits, rows), inp.instance_labels[rows], inp.batch[rows],
                                                  self.opt.embed_dim).items():
                setattr(self, name, loss)
                if name == "ins_loss":
                    self.loss = self.loss + self.opt.loss_weights.embedding_loss * loss
        mask_sigmoid = None if out.mask_scores is None else torch.sigmoid(out.mask_scores).reshape(-1)
        if out.cluster_scores is not None and self._scorer_type and epoch > self.opt.prepare_epoch and self.use_score_net:
            on_mask = bool(self.cal_iou_based_on_mask and epoch > self.cal_iou_based_on_mask_start_epoch)  # reference :592-611
            ious = instance_ious(out.clusters, out.cluster_scores, inp.instance_labels, inp.batch, mask_sigmoid, on_mask,
                                 clusters_csr=out.clusters_csr)
            self.score_loss = instance_iou_loss(ious, out.clusters, out.cluster_scores, inp.instance_labels, inp.batch,
                                                min_iou_threshold=self.opt.min_iou_threshold,
                                                max_iou_threshold=self.opt.max_iou_threshold)
            self.loss = self.loss + self.score_loss * self.opt.loss_weights["score_loss"]
            if mask_sigmoid is not None and self.mask_supervise:  # reference :626-634
                self.mask_loss = mask_loss(ious, out.clusters, mask_sigmoid, inp.instance_labels, inp.batch,
                                           clusters_csr=out.clusters_csr)
                self.loss = self.loss + self.mask_loss * self.opt.loss_weights["mask_loss"]

    def backward(self, epoch):
        self._compute_loss(epoch)
        self.loss.backward()
