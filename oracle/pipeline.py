"""CPU oracle of the WHOLE hot path (eval mode): sparse U-Net -> heads -> grouping -> scorer, driven by a state_dict.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Independent restatement of the
network wiring of torch_points3d/applications/minkowski.py:160-196 and
torch_points3d/modules/MinkowskiEngine/api_modules.py:9-82,235-311 on top of oracle/panoptic_oracle.c -- it does not
import the product package's modules.  "Parity unpinned" vs MinkowskiEngine itself (see panoptic_oracle.c header).
"""
import time

import numpy as np

from . import oracle as O


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


class Maps:
    """coordinate levels + kernel maps for one input (sorted-table lookups in C)."""

    def __init__(self, coords):
        self.levels = {1: np.ascontiguousarray(coords, np.int32)}
        self.maps = {}

    def ensure(self, ts_in, stride):
        ts_out = ts_in * stride
        if ts_out not in self.levels:
            self.levels[ts_out], _ = O.stride_coords(self.levels[ts_in], ts_out)
        return ts_out

    def map(self, ts_from, ts_to, sign):
        key = (ts_from, ts_to, sign)
        if key not in self.maps:
            self.maps[key] = O.kernel_map(self.levels[ts_to], self.levels[ts_from], 3, min(ts_from, ts_to), sign)
        return self.maps[key]


CONV_STATS = {"flops": 0.0, "seconds": 0.0}  # accumulated by every convolution call (bench.py cpu_baseline reads it)


def _fold(sd, p):
    w, b = _np(sd[p + ".bn.weight"]), _np(sd[p + ".bn.bias"])
    m, v = _np(sd[p + ".bn.running_mean"]), _np(sd[p + ".bn.running_var"])
    scale = w / np.sqrt(v + 1e-5)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def _conv_bn(sd, pconv, pbn, maps, x, ts_in, stride, transposed, relu, residual=None, in1=None):
    W = _np(sd[pconv + ".kernel"])
    sign = -1 if transposed else 1
    if W.ndim == 2:
        ts_out, nbr = ts_in, None
    else:
        ts_out = ts_in if stride == 1 else (ts_in // stride if transposed else maps.ensure(ts_in, stride))
        nbr = maps.map(ts_in, ts_out, sign)
    n_out = len(maps.levels[ts_out])
    scale, shift = _fold(sd, pbn)
    t0 = time.perf_counter()
    y = O.spconv_fwd(x, W, nbr, n_out, in1=in1, scale=scale, shift=shift, relu=relu, residual=residual)
    # bookkeeping for the CPU baseline's GFLOP/s (2 flops per pair, input and output channel)
    pairs = n_out if nbr is None else int((nbr >= 0).sum())
    CONV_STATS["flops"] += 2.0 * pairs * W.shape[-2] * W.shape[-1]
    CONV_STATS["seconds"] += time.perf_counter() - t0
    return y, ts_out


def _resblock(sd, p, maps, x, ts, transposed):
    has_ds = (p + ".downsample.0.kernel") in sd
    res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", maps, x, ts, 1, transposed, False)[0] if has_ds else x
    h, _ = _conv_bn(sd, p + ".block.0", p + ".block.1", maps, x, ts, 1, transposed, True)
    y, _ = _conv_bn(sd, p + ".block.3", p + ".block.4", maps, h, ts, 1, transposed, True, residual=res)
    return y


def _level(sd, p, maps, x, ts, stride, transposed, n_blocks, skip=None):
    y, ts = _conv_bn(sd, p + ".conv_in.0", p + ".conv_in.1", maps, x, ts, stride, transposed, True, in1=skip)
    for b in range(n_blocks):
        y = _resblock(sd, "%s.blocks.%d" % (p, b), maps, y, ts, transposed)
    return y, ts


def unet_forward(sd, prefix, down_strides, up_strides, n_blocks, coords, feats):
    """coords int32 [N,4] (b,x,y,z), feats f32 [N,C] -> f32 [N,Cout], row-aligned with the input."""
    maps = Maps(coords)
    x, ts = np.ascontiguousarray(feats, np.float32), 1
    stack = []
    nd = len(down_strides)
    for i in range(nd):
        x, ts = _level(sd, "%s.down_modules.%d" % (prefix, i), maps, x, ts, down_strides[i], False, n_blocks)
        stack.append((x, ts) if i < nd - 1 else None)
    for i in range(len(up_strides)):
        skip = stack.pop()
        x, ts = _level(sd, "%s.up_modules.%d" % (prefix, i), maps, x, ts, up_strides[i], True, n_blocks,
                       skip=None if skip is None else skip[0])
    assert ts == 1
    return x


def encoder_forward(sd, prefix, down_strides, n_blocks, coords, feats):
    """MinkowskiEncoder (applications/minkowski.py:129-156): the down modules, then the innermost GlobalBaseModule -- a
    Linear + BatchNorm + LeakyReLU(0.2) MLP per point and a per-batch-element maximum (core/base_conv/message_passing.py:
    132-151).  Returns f32 [n_batch, C] in batch order."""
    maps = Maps(coords)
    x, ts = np.ascontiguousarray(feats, np.float32), 1
    for i, st in enumerate(down_strides):
        x, ts = _level(sd, "%s.down_modules.%d" % (prefix, i), maps, x, ts, st, False, n_blocks)
    x = mlp_forward(sd, prefix + ".inner_modules.0.nn", x)
    b = maps.levels[ts][:, 0].astype(np.int64)
    out, _ = O.segment_reduce(x, b, int(coords[:, 0].max()) + 1, "max")
    return out


def mlp_forward(sd, p, x, slope=0.2):
    """core/common_modules/base_modules.py:35-45: [Linear(bias) -> BatchNorm1d (eval) -> LeakyReLU(0.2)] per layer."""
    i = 0
    x = np.asarray(x, np.float32)
    while "%s.%d.0.weight" % (p, i) in sd:
        q = "%s.%d" % (p, i)
        y = x.astype(np.float64) @ _np(sd[q + ".0.weight"]).astype(np.float64).T
        if q + ".0.bias" in sd:
            y = y + _np(sd[q + ".0.bias"]).astype(np.float64)
        bn = q + ".1.batch_norm"
        w, b = _np(sd[bn + ".weight"]).astype(np.float64), _np(sd[bn + ".bias"]).astype(np.float64)
        m, v = _np(sd[bn + ".running_mean"]).astype(np.float64), _np(sd[bn + ".running_var"]).astype(np.float64)
        y = (y - m) / np.sqrt(v + 1e-5) * w + b
        x = np.where(y > 0, y, slope * y).astype(np.float32)
        i += 1
    return x


def _head(sd, p, x, log_softmax=False):
    bn = p + ".0.0.1.batch_norm"
    w, b = _np(sd[bn + ".weight"]), _np(sd[bn + ".bias"])
    m, v = _np(sd[bn + ".running_mean"]), _np(sd[bn + ".running_var"])
    scale = (w / np.sqrt(v + 1e-5)).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return O.head_mlp(x, _np(sd[p + ".0.0.0.weight"]), scale, shift, _np(sd[p + ".1.weight"]), _np(sd[p + ".1.bias"]),
                      log_softmax=log_softmax, want_argmax=True)


def nms(ious, scores, threshold):
    """non_max_suppression of structure_3heads.py:6-16.  The reference orders with `scores.argsort()[::-1]` (numpy
    introsort: the order of EQUAL scores is implementation-defined beyond 16 elements); the tie rule is pinned here to the
    stable sort reversed -- descending score, equal scores by descending index -- which is what numpy's insertion-sort
    path gives, and what csrc/pp_nms.hip implements."""
    ixs = np.argsort(scores, kind="stable")[::-1]
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        remove = np.where(ious[i, ixs[1:]] > threshold)[0] + 1
        ixs = np.delete(ixs, remove)
        ixs = np.delete(ixs, 0)
    return pick


def sklearn_meanshift_labels(args):
    """the reference's worker (torch_points3d/utils/meanshift_cluster.py:9-18): one sample, sklearn MeanShift with bin
    seeding.  Module-level so that a spawned multiprocessing.Pool can import it (the reference maps this function over the
    samples of a batch, one process per sample, :96-101)."""
    x, bandwidth = args
    from sklearn.cluster import MeanShift
    return MeanShift(bandwidth=bandwidth, bin_seeding=True).fit(x).labels_


def meanshift_clusters(emb, batch, local_ind, bandwidth, use_sklearn=False, pool=None):
    """cluster_single of torch_points3d/utils/meanshift_cluster.py:72-123 (one MeanShift per batch element; pool: the
    samples are mapped over a multiprocessing.Pool as the reference does, :96-101)."""
    out = []
    samples = [s for s in np.unique(batch) if (batch == s).sum() > 3]
    pooled = None
    if use_sklearn and pool is not None:
        pooled = pool.map(sklearn_meanshift_labels, [(emb[batch == s], bandwidth) for s in samples])
    for j, s in enumerate(samples):
        m = batch == s
        x = emb[m]
        if pooled is not None:
            labels = pooled[j]
        elif use_sklearn:
            labels = sklearn_meanshift_labels((x, bandwidth))
        else:
            labels, _, _ = O.meanshift(x, [0, len(x)], bandwidth)
        li = local_ind[m]
        for l in np.unique(labels):
            if l == -1:
                continue
            out.append(li[labels == l])
    return out


def group(pos, batch, pred, off, emb, opt, stuff_classes, use_sklearn_meanshift=False, timings=None, ms_clusters=None):
    """Proposal generation of PointGroup3heads (PointGroup3heads.py:163-390): cluster_type 1 = region growing on the shifted
    points (nsample 200), 2 = on the raw positions (torch-points-kernels' default nsample 16) then on the shifted points,
    3 = mean shift on the embeddings of the thing points alone, 4 = raw positions then mean shift,
    5 = shifted points then mean shift on the embeddings of the thing points, 6 = raw, shifted, mean shift.  Returns
    (clusters, cluster_type codes) in the reference's order; _cluster2 marks the votes as type 1 only when there are
    position clusters (:208-210).  Pinned on the reference's own functions: tests/golden/proposal_cases.npz."""
    T = {} if timings is None else timings
    ct = int(opt["cluster_type"])
    ignore = [-1] + [int(c) for c in stuff_classes]
    radius = float(opt["cluster_radius_search"])
    pos_cl = []
    if ct in (2, 4, 6):
        pos_cl, _ = O.region_grow(pos, pred, batch, ignore, nsample=16, radius=radius, min_cluster_size=10)
    votes = []
    if ct in (1, 2, 5, 6):
        t0 = time.perf_counter()
        votes, _ = O.region_grow(pos + off, pred, batch, ignore, nsample=200, radius=radius, min_cluster_size=10)
        T["region_grow"] = time.perf_counter() - t0
    ms = []
    if ct in (3, 4, 5, 6):
        t0 = time.perf_counter()
        mask = ~np.isin(pred, ignore)
        # ms_clusters: the mean-shift proposals of an earlier pass on the same inputs (bench.py times the embedding
        # clustering separately, fanned out over processes as the reference does)
        ms = list(ms_clusters) if ms_clusters is not None else meanshift_clusters(
            emb[mask], batch[mask], np.nonzero(mask)[0], float(opt["bandwidth"]), use_sklearn=use_sklearn_meanshift)
        T["meanshift"] = time.perf_counter() - t0
    if ct == 1:
        return list(votes), [0] * len(votes)
    if ct == 2:
        return list(pos_cl) + list(votes), [0] * len(pos_cl) + [1 if len(pos_cl) else 0] * len(votes)
    if ct == 3:  # mean shift on the embeddings alone (PointGroup3heads.py:213-243)
        return ms, [0] * len(ms)
    if ct == 4:  # raw positions (default nsample) + mean shift (:246-289)
        return list(pos_cl) + ms, [0] * len(pos_cl) + [1] * len(ms)
    if ct == 5:
        return list(votes) + ms, [0] * len(votes) + [1] * len(ms)
    if ct == 6:
        return list(pos_cl) + list(votes) + ms, [0] * len(pos_cl) + [1] * len(votes) + [2] * len(ms)
    raise NotImplementedError("cluster_type %d" % ct)


def forward(sd, data, opt, num_classes, stuff_classes, override=None, use_sklearn_meanshift=False, timings=None, ms_clusters=None,
            scorer_type="unet"):
    """Eval forward of PointGroup3heads (setting IV / cluster_type 5 or type 1) on CPU.
    data: dict with pos [N,3], coords [N,3], batch [N], x [N,4].  Returns dict of outputs."""
    T = {} if timings is None else timings
    t0 = time.perf_counter()
    coords4 = np.concatenate([data["batch"][:, None], data["coords"]], 1).astype(np.int32)
    feats = unet_forward(sd, "Backbone", [1, 2, 2, 2, 2, 2, 2], [2, 2, 2, 2, 2, 2, 1], 2, coords4, data["x"])
    T["unet"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sem, pred = _head(sd, "Semantic", feats, log_softmax=True)
    off, _ = _head(sd, "Offset", feats)
    emb, _ = _head(sd, "Embed", feats)
    T["heads"] = time.perf_counter() - t0
    if override is not None:
        pred, off, emb = override
    clusters, ctype = group(data["pos"], data["batch"], pred, off, emb, opt, stuff_classes, use_sklearn_meanshift, T, ms_clusters)
    clusters, ctype = list(clusters), list(ctype)
    scores = cf = None
    if clusters:
        t0 = time.perf_counter()
        pts = np.concatenate(clusters)
        b = np.concatenate([np.full(len(c), i) for i, c in enumerate(clusters)])
        sc_coords = np.concatenate([b[:, None], data["coords"][pts]], 1).astype(np.int32)
        if scorer_type == "MLP":      # PointGroup3heads.py:419-423
            cf, _ = O.segment_reduce(mlp_forward(sd, "ScorerMLP", feats[pts]), b, len(clusters), "max")
        elif scorer_type == "encoder":  # :424-426
            cf = encoder_forward(sd, "ScorerEncoder", [2, 2], 2, sc_coords, feats[pts])
        else:
            sf = unet_forward(sd, "ScorerUnet", [2, 2], [2, 2], 2, sc_coords, feats[pts])
            cf, _ = O.segment_reduce(sf, b, len(clusters), "max")
        w, bb = _np(sd["ScorerHead.0.weight"]), _np(sd["ScorerHead.0.bias"])
        scores = 1.0 / (1.0 + np.exp(-(cf @ w.T + bb)[:, 0]))
        T["scorer"] = time.perf_counter() - t0
    return {"features": feats, "semantic_logits": sem, "offset_logits": off, "embed_logits": emb, "pred": pred,
            "clusters": clusters, "cluster_type": np.asarray(ctype, np.uint8), "cluster_scores": scores,
            "cluster_features": cf}  # per-proposal maximum of the scorer's features: the ScorerHead's input


def instance_labels(out, n_points, batch, nms_threshold=0.3, min_cluster_points=10, min_score=0.5):
    """get_instances + get_cur_ins_pre_label per batch element (tracker :326-337, structure_3heads.py:28-71)."""
    labels = np.full(n_points, -1, np.int32)
    clusters, scores = out["clusters"], out["cluster_scores"]
    if not clusters:
        return labels
    tile = np.asarray([batch[c[0]] for c in clusters])
    for t in np.unique(tile):
        ids = np.nonzero(tile == t)[0]
        cl = [clusters[i] for i in ids]
        inter = O.proposal_intersections(cl, n_points).astype(np.float32)
        sz = np.diag(inter).copy()
        ious = inter / (sz[:, None] + sz[None, :] - inter)
        sc = scores[ids]
        pick = nms(ious, sc, nms_threshold)
        keep = [i for i in pick if sz[i] > min_cluster_points and sc[i] > min_score]
        order = [keep[j] for j in np.argsort(sc[keep], kind="stable")] if keep else []
        for rank, i in enumerate(order):
            labels[cl[i]] = rank
    return labels
