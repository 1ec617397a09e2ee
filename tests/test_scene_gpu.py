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
    for t in range(len(tiles)):
        b = syn.tile_batch(scene, tiles, [t])
        ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(50 + t))
        labels, res, _ = runner.run(b, 1, override=tuple(torch.from_numpy(a).to(dev) for a in ov))
        want = opipe.forward(sd, b, opt, 9, syn.NPM3D_STUFF, override=ov)
        want["cluster_scores"] = res.cluster_scores.cpu().numpy()  # same scores -> NMS / paint order is deterministic
        want_labels = opipe.instance_labels(want, len(b["pos"]), b["batch"])
        # per-tile labels identical after canonicalisation
        import bruteforce as bf
        assert np.array_equal(bf.canon_partition(labels.cpu().numpy()), bf.canon_partition(want_labels))
        asm_gpu.add_block(b["origin_id"], labels.cpu().numpy())
        asm_cpu.add_block(b["origin_id"], want_labels)
    # block merging is a function of the (canonical) per-tile partitions in block order
    assert np.array_equal(bf.canon_partition(asm_gpu.ins_pre), bf.canon_partition(asm_cpu.ins_pre))
    pq_gpu = thing_panoptic_quality(scene.cls, asm_gpu.ins_pre, scene.cls, scene.inst, syn.THING_CLASSES)
    pq_cpu = thing_panoptic_quality(scene.cls, asm_cpu.ins_pre, scene.cls, scene.inst, syn.THING_CLASSES)
    assert abs(pq_gpu["PQ"] - pq_cpu["PQ"]) < 1e-9
    assert pq_gpu["PQ"] > 0.5  # synthetic head statistics are good, so grouping must recover most instances


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
