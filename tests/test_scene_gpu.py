"""GPU: scene-level run (tiles -> model -> per-tile labels -> block merging) through the product path vs the CPU oracle
pipeline, and the PQ of both against the generator's ground truth (north star: PQ within +-0.1 of the reference path;
here the two paths give identical scene labels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_scene_labels_and_pq_match_oracle():
    import bench
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import thing_panoptic_quality
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler, TileRunner
    dev = torch.device("cuda")
    scene, tiles, _ = bench.build_scene(90_000, 2, 0.05, 2022)
    model, cfg, DS = bench.build_model(dev, 0.05)
    runner = TileRunner(model, dev)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    opt = {"cluster_radius_search": cfg.cluster_radius_search, "cluster_type": cfg.cluster_type, "bandwidth": cfg.bandwidth}
    asm_gpu, asm_cpu = SceneAssembler(len(scene.pos), 9), SceneAssembler(len(scene.pos), 9)
    import bruteforce as bf
    n_amb = n_pts = 0
    for t in range(len(tiles)):
        b = syn.tile_batch(scene, tiles, [t])
        ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(50 + t))
        ovd = tuple(torch.from_numpy(a).to(dev) for a in ov)
        _, res0, _ = runner.run(b, 1, override=ovd)
        # NO score substitution: the oracle paints with its OWN scores and NMS.  A random-init ScorerHead squeezes all scores
        # into a ~1e-3 band where rounding decides the paint order, so its logits are spread first (same features, same
        # proposals); points that still hang on a comparison closer than 1e-5 between overlapping twins are counted and bounded
        with bf.spread_scorer_head(model.ScorerHead[0], res0.cluster_scores):
            labels, res, _ = runner.run(b, 1, override=ovd)
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            want = opipe.forward(sd, b, opt, 9, syn.NPM3D_STUFF, override=ov)
        assert bf.scaled_err("tile %d proposal scores (spread)" % t, res.cluster_scores.cpu().numpy(), want["cluster_scores"]) < 1e-4
        want_labels = opipe.instance_labels(want, len(b["pos"]), b["batch"])
        amb = bf.near_tie_points(want["clusters"], want["cluster_scores"], len(b["pos"]), other_scores=res.cluster_scores.cpu().numpy())
        n_amb, n_pts = n_amb + int(amb.sum()), n_pts + len(amb)
        got_labels = labels.cpu().numpy()
        assert np.array_equal(bf.canon_partition(got_labels[~amb]), bf.canon_partition(want_labels[~amb]))
        # the scene assembly below sees the ambiguous points unlabelled in both chains
        got_labels, want_labels = np.where(amb, -1, got_labels), np.where(amb, -1, want_labels)
        asm_gpu.add_block(b["origin_id"], got_labels)
        asm_cpu.add_block(b["origin_id"], want_labels)
    print("scene: %d of %d tile points hang on a score comparison closer than 1e-5 (left unlabelled in both chains)" % (n_amb, n_pts))
    assert n_amb <= 0.05 * n_pts
    # block merging is a function of the (canonical) per-tile partitions in block order
    assert np.array_equal(bf.canon_partition(asm_gpu.ins_pre), bf.canon_partition(asm_cpu.ins_pre))
    pq_gpu = thing_panoptic_quality(scene.cls, asm_gpu.ins_pre, scene.cls, scene.inst, syn.THING_CLASSES)
    pq_cpu = thing_panoptic_quality(scene.cls, asm_cpu.ins_pre, scene.cls, scene.inst, syn.THING_CLASSES)
    assert abs(pq_gpu["PQ"] - pq_cpu["PQ"]) < 1e-9
    assert pq_gpu["PQ"] > 0.5  # synthetic head statistics are good, so grouping must recover most instances


def test_prepared_next_batch_does_not_change_results():
    """TileRunner.run(next_batch=...) builds the next batch's coordinate manager on its own thread and streams during the
    current batch (ME.PreparedCoordinates); the pass that takes it over gives bit-identical results to one that builds its
    own, a pass whose batch is NOT the prepared one ignores it, and the environment switch turns it off."""
    import bench
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd import scene as scene_mod
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    dev = torch.device("cuda")
    scene, tiles, _ = bench.build_scene(120_000, 2, 0.05, 2023)
    model, cfg, DS = bench.build_model(dev, 0.05)
    runner = TileRunner(model, dev)
    batches = []
    for ids in ([0, 1], [2, 3]):
        b = syn.tile_batch(scene, tiles, ids)
        ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(7 + ids[0]))
        batches.append(({k: torch.from_numpy(v).to(dev) for k, v in b.items()}, tuple(torch.from_numpy(a).to(dev) for a in ov)))

    def run(i, nxt=None):
        labels, res, counts = runner.run(batches[i][0], 2, override=batches[i][1], next_batch=None if nxt is None else batches[nxt][0])
        return labels.cpu().numpy(), res.cluster_scores.cpu().numpy(), res.semantic_logits.cpu().numpy(), counts

    for _ in range(2):  # (second round: the map prefetch plan exists, the prepared manager starts its own builder)
        plain = [run(0), run(1)]
        a = run(0, nxt=1)                                   # prepares batch 1 ...
        assert "_prepared_input" in model.Backbone.__dict__
        b = run(1, nxt=0)                                   # ... which this pass takes over, preparing batch 0
        c = run(1)                                          # batch 0 was prepared, batch 1 is run: ignored and dropped
        assert "_prepared_input" not in model.Backbone.__dict__
        for got, want in ((a, plain[0]), (b, plain[1]), (c, plain[1])):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            assert got[3] == want[3]
    scene_mod.INPUT_PREFETCH = False
    try:
        d = run(0, nxt=1)
        assert "_prepared_input" not in model.Backbone.__dict__
        assert np.array_equal(d[0], plain[0][0])
    finally:
        scene_mod.INPUT_PREFETCH = True


@pytest.mark.parametrize("ahead_thread", [False, True])
def test_backbone_ahead_does_not_change_results(ahead_thread):
    """TileRunner(backbone_ahead=True) -- enqueued by the calling thread or (ahead_thread) by a host thread of its own: the NEXT batch's backbone + heads run on a stream of their own beside this batch's grouping
    and scorer front end, the batch after next gets its coordinate manager prepared, the scorer's convolutions wait for the
    backbone ahead.  Over a stream of three different batches (twice round, then with a batch that was NOT the announced one) the
    labels, scores, semantic outputs and counts are those of one batch at a time, bit for bit; the real head outputs (no
    override) cross the streams as well."""
    import bench
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    dev = torch.device("cuda")
    scene, tiles, _ = bench.build_scene(150_000, 3, 0.05, 2024)
    model, cfg, DS = bench.build_model(dev, 0.05)
    batches = []
    for ids in ([0, 1], [2, 3], [4, 5, 6]):
        b = syn.tile_batch(scene, tiles, ids)
        ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(17 + ids[0]))
        batches.append(({k: torch.from_numpy(v).to(dev) for k, v in b.items()}, tuple(torch.from_numpy(a).to(dev) for a in ov), len(ids)))

    def out(r):
        labels, res, counts = r
        return (labels.cpu().numpy(), res.cluster_scores.cpu().numpy(), res.semantic_logits.cpu().numpy(),
                res.embed_logits.cpu().numpy(), res.clusters_csr.points.cpu().numpy(), list(counts))

    serial = TileRunner(model, dev)
    want = [out(serial.run(b, n, override=ov)) for b, ov, n in batches]
    want_own = out(serial.run(batches[1][0], batches[1][2]))   # the network's OWN head outputs feed the grouping
    ahead = TileRunner(model, dev, backbone_ahead=True, ahead_thread=ahead_thread)
    order = [0, 1, 2, 0, 1, 2, 0]
    for j, i in enumerate(order):
        nxt = batches[order[j + 1]][0] if j + 1 < len(order) else None
        nxt2 = batches[order[j + 2]][0] if j + 2 < len(order) else None
        got = out(ahead.run(batches[i][0], batches[i][2], override=batches[i][1], next_batch=nxt, after_next=nxt2))
        assert (ahead._ahead is not None) == (nxt is not None)
        for g, w in zip(got, want[i]):
            assert np.array_equal(g, w) if isinstance(g, np.ndarray) else g == w, (j, i)
    # a batch other than the announced one: the backbone that ran ahead is dropped, this batch runs its own
    got = out(ahead.run(batches[0][0], batches[0][2], override=batches[0][1], next_batch=batches[2][0]))
    got = out(ahead.run(batches[1][0], batches[1][2], next_batch=batches[1][0]))   # (announced 2, given 1; no override)
    for g, w in zip(got, want_own):
        assert np.array_equal(g, w) if isinstance(g, np.ndarray) else g == w
    got = out(ahead.run(batches[1][0], batches[1][2]))                              # takes the backbone launched by the call before
    for g, w in zip(got, want_own):
        assert np.array_equal(g, w) if isinstance(g, np.ndarray) else g == w
    ahead.drain()
    assert ahead._ahead is None and "_before_first_conv" not in model.ScorerUnet.__dict__


def test_tile_batch_on_the_gpu_matches_numpy_collation():
    """voxelise -> cut cylinders -> collate on the GPU (row f1) vs the NumPy generator path on the same raw cloud."""
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd import ops, synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import tile_batch_gpu
    dev = torch.device("cuda")
    rng = np.random.default_rng(8)
    raw, cls, inst = syn.urban_points(120_000, 30.0, rng)
    coords, rep, inv = ops.voxelize(torch.from_numpy(raw).to(dev), 0.05)
    wc, wr, _ = oracle.voxelize(raw, 0.05)
    assert np.array_equal(rep.cpu().numpy(), wr)
    pos_v = torch.from_numpy(raw).to(dev)[rep]                    # one REAL point per voxel
    coords3 = coords[:, 1:].contiguous()
    cen = torch.tensor([[8.0, 8.0], [16.0, 9.0], [22.0, 21.0], [9.0, 22.0]], device=dev)
    got = tile_batch_gpu(pos_v, coords3, 0.05, cen, 7.5)
    # NumPy side: same voxels, cylinders from the oracle, collation from the synthetic generator
    scene = syn.Scene(raw[wr], wc[:, 1:], cls[wr], inst[wr], 0.05, 30.0)
    tiles = oracle.cylinder_tiles(scene.pos, cen.cpu().numpy(), 7.5)
    want = syn.tile_batch(scene, tiles, [0, 1, 2, 3])
    assert np.array_equal(got["origin_id"].cpu().numpy(), want["origin_id"])
    assert np.array_equal(got["batch"].cpu().numpy(), want["batch"])
    assert np.array_equal(got["coords"].cpu().numpy(), want["coords"])
    # the cylinder mean is accumulated in float64 on the GPU and in float32 (pairwise) by NumPy: ~1e-4 m apart
    np.testing.assert_allclose(got["pos"].cpu().numpy(), want["pos"], atol=3e-4)
    np.testing.assert_allclose(got["x"].cpu().numpy(), want["x"], atol=5e-4)


def _cloud(rng, n, dim, spread=40.0):
    """clustered surface-like cloud with a few far outliers"""
    c = rng.uniform(-spread, spread, size=(max(n // 200, 1), dim))
    p = c[rng.integers(0, len(c), n)] + rng.normal(size=(n, dim)) * rng.uniform(0.05, 2.0, size=(n, 1))
    p[: max(n // 500, 1)] += rng.normal(size=(max(n // 500, 1), dim)) * 400
    return p.astype(np.float32)


@pytest.mark.parametrize("dim", [3, 2])
def test_nearest_matches_bruteforce_bit_exact(dim):
    """pp_nearest against the brute-force oracle: indices and float32 squared distances identical, including ties
    (duplicated and lattice points), far outliers, a bounded search and an empty reference set."""
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd import ops
    rng = np.random.default_rng(5 + dim)
    dev = torch.device("cuda")
    for n_ref, n_q, cell in [(3000, 6000, 0.5), (1, 500, 1.0), (5000, 4000, 0.05), (2000, 3000, 7.0), (40, 2000, 0.3)]:
        ref = _cloud(rng, n_ref, dim)
        q = np.concatenate([_cloud(rng, n_q // 2, dim), ref[rng.integers(0, n_ref, n_q // 2)] +
                            rng.normal(size=(n_q // 2, dim)).astype(np.float32) * 0.02])
        ref[n_ref // 2:n_ref // 2 + 20] = ref[:20]                    # duplicates: smallest index must win
        for max_dist in (0.0, 1.0):
            want_i, want_d = oracle.nearest(ref, q, max_dist)
            got_i, got_d = ops.nearest(torch.from_numpy(ref).to(dev), torch.from_numpy(q).to(dev), cell, max_dist)
            assert np.array_equal(got_i.cpu().numpy(), want_i)
            assert np.array_equal(got_d.cpu().numpy(), want_d)
    # lattice: many exact ties
    g = np.stack(np.meshgrid(*[np.arange(12)] * dim, indexing="ij"), -1).reshape(-1, dim).astype(np.float32)
    q = (g[rng.integers(0, len(g), 3000)] + rng.choice([0.0, 0.5], size=(3000, dim))).astype(np.float32)
    want_i, want_d = oracle.nearest(g, q)
    got_i, got_d = ops.nearest(torch.from_numpy(g).to(dev), torch.from_numpy(q).to(dev), 1.0)
    assert np.array_equal(got_i.cpu().numpy(), want_i) and np.array_equal(got_d.cpu().numpy(), want_d)
    got_i, got_d = ops.nearest(torch.from_numpy(g[:0]).to(dev), torch.from_numpy(q).to(dev), 1.0)
    assert (got_i == -1).all() and torch.isinf(got_d).all()
    with pytest.raises(Exception):
        ops.nearest(torch.from_numpy(g).to(dev) * 1e9, torch.from_numpy(q).to(dev), 1.0)


def test_grid_cylinder_tiles_match_reference_golden():
    """scene.grid_cylinder_tiles (device PCA grid + cylinders + centre labels) against the output of the reference's
    own GridCylinderSampling (golden fixture, sklearn >= 1.5 sign rule = "v") and, for the pinned-sklearn sign rule
    ("u"), against the oracle restatement."""
    import os
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd import scene as sc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_cylinder_cases.npz"))
    dev = torch.device("cuda")
    pos, radius, grid = z["pos"], float(z["radius"]), float(z["grid_size"])
    tiles, cen, lab = sc.grid_cylinder_tiles(torch.from_numpy(pos).to(dev), radius, grid,
                                             labels=torch.from_numpy(z["y"]).to(dev), svd_flip="v")
    off = z["offsets"]
    assert tiles.n == len(off) - 1
    # the reference's PCA runs in float32 inside sklearn (axes good to ~4e-5 rad => centres good to a few mm over this
    # 100 m strip); the device path reduces in float64.  Tolerance: 2 cm on 8 m spacing; memberships may differ only for
    # points within that distance of a cylinder wall.
    np.testing.assert_allclose(cen.cpu().numpy(), z["centres"], atol=2e-2)
    got_off, got_pts = tiles.offsets.cpu().numpy(), tiles.points.cpu().numpy()
    differing = 0
    for k in range(tiles.n):
        got = got_pts[got_off[k]:got_off[k + 1]]
        assert np.all(np.diff(got) > 0)
        sym = np.setxor1d(got, z["origin"][off[k]:off[k + 1]])
        differing += len(sym)
        if len(sym):
            d = np.linalg.norm(pos[sym, :2] - z["centres"][k], axis=1)
            assert np.all(np.abs(d - radius) < 3e-2)
    assert differing <= 0.002 * len(z["origin"])
    assert (lab.cpu().numpy() != z["centre_label"]).sum() <= 1
    want = oracle.grid_cylinder_centres(pos, grid, u_based=True)
    got = sc.grid_cylinder_centres(torch.from_numpy(pos).to(dev), grid, svd_flip="u").cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-2)


def test_back_project_matches_oracle():
    """full-resolution back-projection (nearest predicted point, stuff / distance / size filters) vs the oracle."""
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd import scene as sc
    rng = np.random.default_rng(8)
    dev = torch.device("cuda")
    n = 12000
    pos = _cloud(rng, n, 3, spread=15.0)
    sub = rng.random(n) < 0.3                                   # the sub-sampled cloud the cylinders saw
    votes = np.zeros((n, 9), np.float32)
    votes[sub] = rng.random((int(sub.sum()), 9)).astype(np.float32)
    count = sub.astype(np.int64) * rng.integers(1, 4, n)
    ins = np.full(n, -1, np.int64)
    has = sub & (rng.random(n) < 0.6)
    ins[has] = rng.integers(0, 150, int(has.sum()))
    want_sem, want_ins = oracle.back_project(pos, votes, count, ins, [0, 1, 5])
    t = lambda a: torch.from_numpy(a).to(dev)
    got_sem, got_ins = sc.back_project(t(pos), t(votes), t(count), t(ins), [0, 1, 5], cell=0.5)
    assert np.array_equal(got_sem.cpu().numpy(), want_sem)
    assert np.array_equal(got_ins.cpu().numpy(), want_ins)
    assert (want_ins == -1).any() and (want_ins >= 0).any()
