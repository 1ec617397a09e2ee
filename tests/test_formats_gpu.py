"""GPU, SURVEY.md §8 row f4: the on-disk formats THROUGH the device path.
  * a checkpoint file in the reference trainer's layout (torch_points3d/metrics/model_checkpoint.py:38-52: `models.latest`
    state_dict, `run_config`, ...) written from one model loads key for key (strict) into a second GPU model, whose outputs
    on the device path are then bit-identical to the source model's; the keys / shapes in the file are the reference's own
    (tests/golden/structure_fixture.json, derived from the reference's api_modules.py + applications/minkowski.py);
  * an input cloud as a PLY with the reference's field names (datasets/segmentation/npm3d.py:76-93) -> `pp_voxelize` ->
    cylinders -> model -> scene assembly -> `back_project` -> evaluation PLY (`preds` / `gt` int16, datasets/panoptic/
    npm3d.py:70-85), compared with the same chain through oracle/ (CPU restatement)."""
import json
import os

import numpy as np
import pytest
import torch

import bruteforce as bf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(model, b, ov, dev):
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    n_tiles = int(b["batch"].max()) + 1
    labels, res, counts = TileRunner(model, dev).run(b, n_tiles, override=tuple(torch.from_numpy(a).to(dev) for a in ov))
    return labels, res, counts


def test_reference_layout_checkpoint_reproduces_the_model_on_the_device(tmp_path):
    import bench
    from panopticsegforlargescalepointcloud_amd import io as pio, synthetic as syn
    dev = torch.device("cuda")
    src, cfg, DS = bench.build_model(dev, 0.05)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():  # a trained model has non-trivial BatchNorm statistics: randomise them so that they matter
        for name, buf in src.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g).to(dev) * 0.1)
            elif name.endswith("running_var"):
                buf.copy_((torch.rand(buf.shape, generator=g) + 0.5).to(dev))
    ck = str(tmp_path / "PointGroup-PAPER.pt")
    pio.save_checkpoint(ck, src, weight_name="latest", run_config={"model_name": "PointGroup-PAPER", "data": {"grid_size": 0.05}})
    raw = torch.load(ck, weights_only=False)
    assert set(raw) >= {"models", "optimizer", "schedulers", "stats", "run_config", "dataset_properties"}
    sd = raw["models"]["latest"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "structure_fixture.json")))["networks"]
    for part in ("Backbone", "ScorerUnet"):  # the file holds the reference's own parameter names and shapes
        have = {k[len(part) + 1:]: list(v.shape) for k, v in sd.items() if k.startswith(part + ".")}
        assert have == fx[part]
    torch.manual_seed(1234)  # a second model with different weights ...
    import copy
    from panopticsegforlargescalepointcloud_amd import panoptic
    dst = panoptic.PointGroup3heads(copy.deepcopy(cfg), "dummy", DS, None).to(dev).eval()
    assert not torch.equal(dst.Backbone.state_dict()[next(iter(dst.Backbone.state_dict()))],
                           src.Backbone.state_dict()[next(iter(src.Backbone.state_dict()))])
    missing, unexpected = pio.load_checkpoint(ck, dst, strict=True)
    assert not missing and not unexpected
    bad = dict(sd)
    bad.pop(next(k for k in bad if k.endswith(".kernel")))
    torch.save({"models": {"latest": bad}}, str(tmp_path / "bad.pt"))
    with pytest.raises(RuntimeError):  # strict: a missing kernel is an error, not a silent random layer
        pio.load_checkpoint(str(tmp_path / "bad.pt"), dst, strict=True)
    # ... reproduces the source model bit for bit on the device path
    scene, tiles, _ = bench.build_scene(50_000, 2, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, [0, 2])
    ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(7))
    l0, r0, c0 = _run(src, b, ov, dev)
    l1, r1, c1 = _run(dst, b, ov, dev)
    assert torch.equal(r0.semantic_logits, r1.semantic_logits) and torch.equal(r0.embed_logits, r1.embed_logits)
    assert torch.equal(r0.offset_logits, r1.offset_logits) and torch.equal(r0.cluster_scores, r1.cluster_scores)
    assert torch.equal(l0, l1) and c0 == c1 and r0.clusters_csr.n > 2


def test_ply_to_device_path_to_eval_ply_matches_the_oracle_chain(tmp_path, oracle):
    import bench
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import io as pio, ops, scene as sc, synthetic as syn
    dev = torch.device("cuda")
    voxel, radius, extent = 0.05, 4.5, 15.0   # sized so that the oracle's chain (CPU) takes ~40 s
    # ---- the input file: reference field names, float scalar fields as CloudCompare writes them
    raw, cls, inst = syn.urban_points(70_000, extent, np.random.default_rng(8))
    src = pio.write_ply(str(tmp_path / "scene"), [raw, (cls + 1).astype(np.float32), (inst - 1).astype(np.float32)],
                        ["x", "y", "z", "scalar_class", "scalar_label"])
    xyz, sem_gt, ins_gt = pio.read_npm3d(src)
    assert np.array_equal(xyz.numpy(), raw) and np.array_equal(sem_gt.numpy(), cls) and np.array_equal(ins_gt.numpy(), inst)
    n_full = len(raw)
    model, cfg, DS = bench.build_model(dev, voxel)
    opt = {"cluster_radius_search": cfg.cluster_radius_search, "cluster_type": cfg.cluster_type, "bandwidth": cfg.bandwidth}
    cen = np.array([[4.5, 4.5], [10.5, 5.0], [7.5, 10.5]], np.float32)
    # ---- device chain: voxelise -> cylinders -> batch -> model -> assembly -> back-projection
    xyz_d = xyz.to(dev)
    coords, rep, _ = ops.voxelize(xyz_d, voxel)
    pos_v = xyz_d[rep]
    batch = sc.tile_batch_gpu(pos_v, coords[:, 1:].contiguous(), voxel, torch.from_numpy(cen).to(dev), radius)
    scene = syn.Scene(pos_v.cpu().numpy(), coords[:, 1:].cpu().numpy(), cls[rep.cpu().numpy()], inst[rep.cpu().numpy()], voxel, extent)
    origin = batch["origin_id"].cpu().numpy()
    ov = syn.synthetic_head_outputs(scene, origin, 0.0, np.random.default_rng(21))
    dev_batch = {k: batch[k] for k in ("pos", "coords", "batch", "x")}
    _, res0, _ = _run(model, dev_batch, ov, dev)
    # NO score substitution below: the oracle chain paints with its OWN scores and NMS.  A random-init ScorerHead squeezes
    # every score into a ~1e-3 band where float rounding decides the paint order, so its logits are spread first (same
    # features, same proposals -- what test_model_gpu.py::test_instance_labels_match_oracle_without_score_substitution does)
    spread = bf.spread_scorer_head(model.ScorerHead[0], res0.cluster_scores)
    spread.__enter__()
    labels, res, counts = _run(model, dev_batch, ov, dev)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    spread.__exit__(None, None, None)
    bt = batch["batch"]
    # ---- the same chain through the oracle
    wc, wr, _ = oracle.voxelize(raw, voxel)
    assert np.array_equal(rep.cpu().numpy(), wr) and np.array_equal(coords.cpu().numpy(), wc)
    tiles = oracle.cylinder_tiles(scene.pos, cen, radius)
    asm_cpu = sc.SceneAssembler(n_full, DS.num_classes)
    gaps = []
    # the batch as the device collated it (tile_batch_gpu centres with float64 means, NumPy's tile_batch with float32 ones:
    # 1e-4 m apart, checked in test_scene_gpu.py) -- both chains see the same numbers from here on
    host = {k: batch[k].cpu().numpy() for k in ("pos", "coords", "batch", "x")}
    for t in range(len(cen)):
        assert np.array_equal(tiles[t], origin[host["batch"] == t])      # same cylinder membership
    want = opipe.forward(sd, host, opt, DS.num_classes, syn.NPM3D_STUFF, override=ov)
    got_cl = [c.cpu().numpy() for c in res.clusters_csr.to_list()]
    assert len(got_cl) == len(want["clusters"]) and all(np.array_equal(g, np.sort(w)) for g, w in zip(got_cl, want["clusters"]))
    gaps = [bf.scaled_err("semantic log-probs", res.semantic_logits.cpu().numpy(), want["semantic_logits"]),
            bf.scaled_err("proposal scores", res.cluster_scores.cpu().numpy(), want["cluster_scores"])]
    # NMS / painting with the ORACLE'S OWN scores (spread head, see above).  Points whose label hangs on a comparison closer than
    # 1e-5 between two overlapping twin proposals are counted, bounded and left unlabelled in BOTH chains
    want_labels = opipe.instance_labels(want, len(host["pos"]), host["batch"])
    amb = bf.near_tie_points(want["clusters"], want["cluster_scores"], len(host["pos"]), other_scores=res.cluster_scores.cpu().numpy())
    print("file chain: %d of %d tile points hang on a score near-tie" % (amb.sum(), len(amb)))
    assert amb.mean() <= 0.05
    amb_d = torch.from_numpy(amb).to(dev)
    labels = torch.where(amb_d, torch.full_like(labels, -1), labels)
    want_labels = np.where(amb, -1, want_labels)
    asm = sc.SceneAssemblerGPU(n_full, DS.num_classes, dev)
    for t in range(len(cen)):  # block order = tile order; origin ids of the FULL cloud = the voxel's representative point
        m = bt == t
        asm.add_block(rep[batch["origin_id"][m]], labels[m], res.semantic_logits[m])
    asm.finish()
    sem_full, ins_full = sc.back_project(xyz_d, asm.votes, asm.prediction_count, asm.ins_pre, syn.NPM3D_STUFF, cell=4 * voxel)
    out_sem = pio.to_eval_ply(raw, sem_full.cpu().numpy(), cls, str(tmp_path / "Semantic_results_forEval"))
    out_ins = pio.to_eval_ply(raw, ins_full.cpu().numpy(), inst, str(tmp_path / "Instance_results_forEval"))
    for t in range(len(cen)):
        sel = host["batch"] == t
        assert np.array_equal(bf.canon_partition(labels[bt == t].cpu().numpy()), bf.canon_partition(want_labels[sel]))
        asm_cpu.add_block(wr[origin[sel]], want_labels[sel], want["semantic_logits"][sel])
    assert max(gaps) < 1e-4
    want_sem, want_ins = oracle.back_project(raw, asm_cpu.votes, asm_cpu.prediction_count, asm_cpu.ins_pre, list(syn.NPM3D_STUFF))
    # ---- the files hold what the oracle chain computes.  The semantic argmax of a random-init network is decided by vote
    # gaps that can be as small as the two paths' float rounding: points whose two best classes are closer than 1e-3 in the
    # oracle's votes are left out of the comparison (a handful), everywhere else the files must agree exactly.
    d_sem, d_ins = pio.read_ply(out_sem), pio.read_ply(out_ins)
    assert d_sem.dtype.names == ("x", "y", "z", "preds", "gt") and d_sem["preds"].dtype == np.int16
    assert np.array_equal(np.stack([d_sem["x"], d_sem["y"], d_sem["z"]], 1), raw)
    assert np.array_equal(d_sem["gt"], cls.astype(np.int16)) and np.array_equal(d_ins["gt"], inst.astype(np.int16))
    has = asm_cpu.prediction_count > 0
    j, _ = oracle.nearest(raw[has], raw)
    v = np.sort(asm_cpu.votes[has][j], axis=1)
    clear = (v[:, -1] - v[:, -2]) > 1e-3
    assert clear.mean() > 0.99
    assert np.array_equal(d_sem["preds"][clear], want_sem[clear].astype(np.int16))
    same_sem = d_sem["preds"] == want_sem.astype(np.int16)
    # instance ids after canonicalisation (block merging numbers instances by running maximum in both chains: equal here,
    # but only the partition is contractual), on the points whose semantic class agrees
    got_i, want_i = d_ins["preds"].astype(np.int64)[same_sem], want_ins[same_sem]
    assert np.array_equal(bf.canon_partition(got_i), bf.canon_partition(want_i))
    assert (want_i >= 0).sum() > 1000 and len(np.unique(want_i)) > 5
