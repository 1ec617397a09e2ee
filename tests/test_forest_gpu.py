"""GPU: BASELINE.json config C3 -- FOR-instance-like forest tile (10 cm voxels, two semantic classes, tree = thing),
offset + embedding dual clustering (cluster_type 5) through the product path vs the CPU oracle pipeline."""
import copy

import numpy as np
import pytest
import torch

import bruteforce as bf

pytestmark = pytest.mark.gpu


def test_forest_tile_dual_clustering_matches_oracle():
    import bench
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import panoptic, synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    dev = torch.device("cuda")

    class DS:
        feature_dimension = 4
        num_classes = syn.FOR_NUM_CLASSES
        stuff_classes = torch.tensor(syn.FOR_STUFF)
    _, cfg, _ = bench.build_model(dev, 0.10)
    cfg = copy.deepcopy(cfg)
    torch.manual_seed(11)
    model = panoptic.PointGroup3heads(cfg, "dummy", DS, None)
    with torch.no_grad():
        model.ScorerHead[0].bias.fill_(1.0)
    model = model.to(dev).eval()
    assert model.Semantic[1].weight.shape[0] == 2
    scene = syn.forest_scene(70_000, 0.10, 2022)
    tiles, radius = syn.cylinder_tiles(scene, 2)
    b = syn.tile_batch(scene, tiles, [0, 1, 2])
    assert scene.n_inst > 5 and (scene.cls[b["origin_id"]] == 1).mean() > 0.2
    ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(6), offset_sigma=0.1)
    labels, res, counts = TileRunner(model, dev).run(b, 3, override=tuple(torch.from_numpy(a).to(dev) for a in ov))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    opt = {"cluster_radius_search": cfg.cluster_radius_search, "cluster_type": cfg.cluster_type, "bandwidth": cfg.bandwidth}
    want = opipe.forward(sd, b, opt, syn.FOR_NUM_CLASSES, syn.FOR_STUFF, override=ov)
    got = [c.cpu().numpy() for c in res.clusters_csr.to_list()]
    assert len(got) == len(want["clusters"]) and len(got) >= 6
    for g, w in zip(got, want["clusters"]):
        assert np.array_equal(g, np.sort(w))
    assert np.array_equal(res.cluster_type.cpu().numpy(), want["cluster_type"]) and set(want["cluster_type"].tolist()) == {0, 1}
    assert bf.scaled_err("forest semantic log-probs", res.semantic_logits.cpu().numpy(), want["semantic_logits"]) < 1e-4
    assert bf.scaled_err("forest proposal scores", res.cluster_scores.cpu().numpy(), want["cluster_scores"]) < 1e-4
    # instance labels against the oracle's OWN scores and NMS (no substitution), with the scorer head's logits spread
    with bf.spread_scorer_head(model.ScorerHead[0], res.cluster_scores):
        labels, res, counts = TileRunner(model, dev).run(b, 3, override=tuple(torch.from_numpy(a).to(dev) for a in ov))
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        want = opipe.forward(sd, b, opt, syn.FOR_NUM_CLASSES, syn.FOR_STUFF, override=ov)
    assert want["cluster_scores"].max() - want["cluster_scores"].min() > 0.2
    assert bf.scaled_err("forest scores (spread)", res.cluster_scores.cpu().numpy(), want["cluster_scores"]) < 1e-4
    want_labels = opipe.instance_labels(want, len(b["pos"]), b["batch"])
    # everywhere except where a near-tie between two overlapping proposals decides (dual clustering proposes every tree
    # twice, with nearly the same points: their max-pooled scores agree to the last bits)
    amb = bf.near_tie_points(want["clusters"], want["cluster_scores"], len(b["pos"]))
    print("forest: %d of %d points hang on a score comparison closer than 1e-5" % (amb.sum(), len(amb)))
    assert amb.mean() < 0.2
    for t in range(3):
        m = (b["batch"] == t) & ~amb
        assert np.array_equal(bf.canon_partition(labels.cpu().numpy()[m]), bf.canon_partition(want_labels[m]))
    assert sum(counts) > 0
