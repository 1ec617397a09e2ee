"""GPU: the reference's two-head model classes (README settings I-III + the HDBSCAN proposal generator) on the product
path; proposals are checked against the oracle's grouping functions applied to the same head outputs."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cls_name, n_points=60_000, **over):
    import bench
    from panopticsegforlargescalepointcloud_amd import panoptic, synthetic as syn
    from panopticsegforlargescalepointcloud_amd.applications import Data
    dev = torch.device("cuda")
    _, cfg, DS = bench.build_model(dev, 0.05)
    cfg = copy.deepcopy(cfg)
    for k, v in over.items():
        cfg[k] = v
    torch.manual_seed(3)
    model = getattr(panoptic, cls_name)(cfg, "dummy", DS, None).to(dev).eval()
    scene, tiles, _ = bench.build_scene(n_points, 2, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, [0, 1])
    data = Data(pos=torch.from_numpy(b["pos"]), coords=torch.from_numpy(b["coords"]), x=torch.from_numpy(b["x"]),
                batch=torch.from_numpy(b["batch"]))
    return model, cfg, scene, b, data, dev


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), np.sort(w))


def test_pointgroup_settings_ii_iii(oracle):
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    for ct in (1, 2):
        model, cfg, scene, b, data, dev = _setup("PointGroup", cluster_type=ct)
        assert not any(k.startswith("Embed.") for k in model.state_dict())
        model.set_input(data, dev)
        with torch.no_grad():
            feats, sem, off, emb, pred = model.backbone_and_heads()
            assert emb is None
            # random-init heads predict nothing useful: substitute generator statistics for the grouping stage
            cls, off_np, _ = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(4))
            pred = torch.from_numpy(cls).to(dev)
            res = model.group_and_score(100, feats, sem, torch.from_numpy(off_np).to(dev), None, pred)
        stuff = np.concatenate([[-1], syn.NPM3D_STUFF])
        shifted, _ = oracle.region_grow(b["pos"] + off_np, cls, b["batch"], stuff, 200, cfg.cluster_radius_search, 10)
        if ct == 1:
            want, types = shifted, [0] * len(shifted)
        else:
            raw, _ = oracle.region_grow(b["pos"], cls, b["batch"], stuff, 16, cfg.cluster_radius_search, 10)
            # (the reference marks the votes as type 1 only when there are position clusters: pointgroup.py:183-184)
            want, types = raw + shifted, [0] * len(raw) + [1 if len(raw) else 0] * len(shifted)
        _same(res.clusters_csr.to_list(), want)
        assert res.cluster_type.cpu().tolist() == types
        assert res.embed_logits is None and res.cluster_scores.shape[0] == len(want)


def test_pointgroupembed_setting_i_and_hdbscan(oracle):
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.utils import hdbscan_cluster as hc
    for ct in (7, 1, 14):
        model, cfg, scene, b, data, dev = _setup("PointGroupEmbed", cluster_type=ct, use_score_net=False)
        assert not any(k.startswith("Offset.") for k in model.state_dict())
        model.set_input(data, dev)
        with torch.no_grad():
            feats, sem, off, emb, pred = model.backbone_and_heads()
            assert off is None
            cls, _, emb_np = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(4))
            res = model.group_and_score(-1, feats, sem, None, torch.from_numpy(emb_np).to(dev), torch.from_numpy(cls).to(dev))
        thing = ~np.isin(cls, np.concatenate([[-1], syn.NPM3D_STUFF]))
        local = np.nonzero(thing)[0]
        bt = b["batch"][thing]
        offs = [0] + np.cumsum(np.bincount(bt, minlength=2)).tolist()

        def lists(labels, ncl):
            out = []
            for s in range(len(offs) - 1):
                seg = labels[offs[s]: offs[s + 1]]
                out += [local[offs[s]: offs[s + 1]][seg == l] for l in range(ncl[s]) if np.any(seg == l)]
            return out
        if ct == 7:
            want = lists(*oracle.meanshift(emb_np[thing], offs, cfg.bandwidth)[:2])
            types = [0] * len(want)
        elif ct == 14:  # HDBSCAN on the embeddings alone (pointgroupembed.py:683-710)
            want = lists(*oracle.hdbscan(emb_np[thing], offs, 15, 5, 0.006, hc.COUNT_SELF))
            types = [0] * len(want)
            assert len(want) > 0
        else:
            xyz = lists(*oracle.hdbscan(b["pos"][thing], offs, 15, 5, 0.006, hc.COUNT_SELF))
            em = lists(*oracle.hdbscan(emb_np[thing], offs, 15, 5, 0.006, hc.COUNT_SELF))
            want, types = xyz + em, [0] * len(xyz) + [1] * len(em)
            assert len(xyz) > 0 and len(em) > 0
        _same(res.clusters_csr.to_list(), want)
        assert res.cluster_type.cpu().tolist() == types
        assert res.cluster_scores is None            # no ScoreNet: get_instances hands back every proposal
        ids, clusters = res._replace(clusters=res.clusters_csr.to_list()).get_instances()
        assert ids is None and len(clusters) == len(want)


@pytest.mark.parametrize("ct", [2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 15, 16])
def test_pointgroupembed_other_cluster_types(oracle, ct):
    """cluster_type 2-6, 8-13, 15, 16 of pointgroupembed.py (:258-681, :712-783), each restated here from the reference's
    function as a union of oracle primitives; the random feature subsets replay numpy's and torch's global generators in
    the order the reference consumes them.  (9, 10, 12, 15 go through meanshift_cluster.cluster_loop, which raises
    TypeError in the reference -- checked against the restatement with opt.bandwidth.)"""
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.utils import hdbscan_cluster as hc
    model, cfg, scene, b, data, dev = _setup("PointGroupEmbed", n_points=24_000, cluster_type=ct, use_score_net=False)
    model.set_input(data, dev)
    cls, _, emb_np = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(4))
    with torch.no_grad():
        feats, sem, off, emb, pred = model.backbone_and_heads()
        np.random.seed(11)
        torch.manual_seed(11)
        res = model.group_and_score(-1, feats, sem, None, torch.from_numpy(emb_np).to(dev), torch.from_numpy(cls).to(dev))
    stuff = np.concatenate([[-1], syn.NPM3D_STUFF])
    thing = ~np.isin(cls, stuff)
    local = np.nonzero(thing)[0]
    bt = b["batch"][thing]
    offs = [0] + np.cumsum(np.bincount(bt, minlength=2)).tolist()
    xyz, em = b["pos"][thing], emb_np[thing]
    both = np.concatenate([xyz, em], 1)

    def lists(labels, ncl):
        out = []
        for s in range(len(offs) - 1):
            seg = labels[offs[s]: offs[s + 1]]
            out += [local[offs[s]: offs[s + 1]][seg == l] for l in range(ncl[s]) if np.any(seg == l)]
        return out

    def H(x, t):
        got = lists(*oracle.hdbscan(x, offs, 15, 5, 0.006, hc.COUNT_SELF))
        return got, [t] * len(got)

    def M(x, t):
        got = lists(*oracle.meanshift(x, offs, cfg.bandwidth)[:2])
        return got, [t] * len(got)

    def loop(x, sizes, one):
        out, types = [], []
        for i, k in enumerate(sizes):
            cols = torch.multinomial(torch.ones(x.shape[1]), int(k), replacement=False).numpy()
            got, _ = one(np.ascontiguousarray(x[:, cols]), i)
            out += got
            types += [i] * len(got)
        return out, types

    HL = lambda x, lo, hi, n: loop(x, np.random.randint(low=lo, high=hi + 1, size=n), H)   # noqa: E731
    HF = lambda x, n: loop(x, [5] * n, H)                                                   # noqa: E731
    ML = lambda x, n: loop(x, [5] * n, M)                                                   # noqa: E731

    def R(t):
        got, _ = oracle.region_grow(b["pos"], cls, b["batch"], stuff, 16, cfg.cluster_radius_search, 10)
        return got, [t] * len(got)

    np.random.seed(11)
    torch.manual_seed(11)
    if ct == 2:
        parts = [HL(both, 3, 5, 9), H(em, 9)]
    elif ct == 3:
        parts = [HL(both, 3, 5, 9), H(xyz, 9)]
    elif ct == 4:
        parts = [HL(both, 3, 5, 8), H(em, 8), H(xyz, 9)]
    elif ct == 5:
        parts = [HL(both, 3, 5, 10)]
    elif ct == 6:
        parts = [HL(em, 2, 5, 6)]
    elif ct == 8:
        parts = [R(0), M(em, 1)]
    elif ct == 9:
        parts = [R(0), ML(em, 10)]
    elif ct == 10:
        parts = [ML(em, 6)]
    elif ct == 11:
        parts = [HF(em, 6)]
    elif ct == 12:
        parts = [R(6), ML(em, 6)]
    elif ct == 13:
        parts = [HF(em, 6), H(xyz, 6)]
    elif ct == 15:
        parts = [ML(em, 6), H(em, 6)]
    else:
        hl = HL(em, 2, 5, 6)          # drawn first (:770), listed second (:777-780)
        parts = [M(em, 6), hl]
    want = [c for p in parts for c in p[0]]
    types = [t for p in parts for t in p[1]]
    if ct == 12:                      # :638-641: the region-growing proposals come first, their types last
        types = parts[1][1] + parts[0][1]
    assert len(want) > 0
    _same(res.clusters_csr.to_list(), want)
    assert res.cluster_type.cpu().tolist() == types


def test_cluster_functions_match_the_reference_model():
    """PointGroup3heads._cluster / _cluster2 / _cluster3 / _cluster4 / _cluster5 / _cluster6 on the device vs the proposals the reference's OWN
    functions produce on the same inputs (tests/golden/proposal_cases.npz: the reference's methods + its mean-shift module
    executed by make_golden.py, torch-points-kernels' region_grow stood in by the CPU oracle): same proposals, same order,
    same cluster_type codes -- incl. the case where the raw positions give no cluster and _cluster2 labels the votes 0."""
    import bench
    from panopticsegforlargescalepointcloud_amd.applications import Data
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proposal_cases.npz"))
    model, cfg, DS = bench.build_model(torch.device("cuda"), 0.2)
    for name in z["names"].tolist():
        pos, off, emb, pred, batch = (torch.from_numpy(z["%s_%s" % (k, name)]).cuda() for k in ("pos", "off", "emb", "pred", "batch"))
        n = pos.shape[0]
        coords = torch.stack([torch.arange(n), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1).cuda()
        model.set_input(Data(pos=pos, coords=coords, batch=batch, x=torch.zeros((n, 4), device="cuda")), torch.device("cuda"))
        model.opt.cluster_radius_search = float(z["radius_" + name])
        model.opt.bandwidth = float(z["bandwidth_" + name])
        for fn in ("_cluster", "_cluster2", "_cluster3", "_cluster4", "_cluster5", "_cluster6"):
            with torch.no_grad():
                csr, types = getattr(model, fn)(pred, off, emb)
            tag = name + fn
            offs, pts = z["offsets_" + tag], z["points_" + tag]
            got = [c.cpu().numpy() for c in csr.to_list()]
            assert len(got) == len(offs) - 1, tag
            for i, c in enumerate(got):
                assert np.array_equal(c, np.sort(pts[offs[i]: offs[i + 1]])), (tag, i)
            assert np.array_equal(types.cpu().numpy(), z["types_" + tag]), tag


@pytest.mark.parametrize("ct", [1, 4, 8, 12, 16])
def test_pointgroupembed_without_thing_points(ct):
    """every primitive of the recipes on an empty selection (all points predicted as stuff): no proposals, no error"""
    model, cfg, scene, b, data, dev = _setup("PointGroupEmbed", n_points=12_000, cluster_type=ct, use_score_net=False)
    model.set_input(data, dev)
    with torch.no_grad():
        feats, sem, off, emb, pred = model.backbone_and_heads()
        res = model.group_and_score(-1, feats, sem, None, emb, torch.zeros_like(pred))   # class 0 = ground (stuff)
    assert res.clusters_csr.n == 0 and res.cluster_type.numel() == 0
