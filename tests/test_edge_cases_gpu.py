"""GPU: degenerate inputs through the product path -- no thing points (no proposals), a single voxel, a tile with one
instance only, coordinates at the ends of the 16-bit range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    import bench
    return bench.build_model(torch.device("cuda"), 0.05)


def test_no_thing_points_gives_no_proposals():
    import bench
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    dev = torch.device("cuda")
    model, cfg, DS = _model()
    scene, tiles, _ = bench.build_scene(30_000, 2, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, [0, 1])
    n = len(b["pos"])
    cls = np.zeros(n, np.int64)                                   # everything "ground" (stuff)
    ov = (cls, np.zeros((n, 3), np.float32), np.zeros((n, 5), np.float32))
    labels, res, counts = TileRunner(model, dev).run(b, 2, override=tuple(torch.from_numpy(a).to(dev) for a in ov))
    assert res.clusters_csr.n == 0 and res.cluster_scores is None
    assert bool((labels == -1).all()) and sum(counts) == 0
    ids, clusters = res._replace(clusters=[]).get_instances()
    assert ids == [] and clusters == []


def test_single_voxel_and_tiny_inputs_run_through_the_unet():
    from panopticsegforlargescalepointcloud_amd.applications import Data
    dev = torch.device("cuda")
    model, cfg, DS = _model()
    for n in (1, 2, 17):
        coords = torch.stack([torch.arange(n), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1).int()
        data = Data(pos=coords.float() * 0.05, coords=coords, x=torch.randn(n, 4), batch=torch.zeros(n, dtype=torch.long))
        model.set_input(data, dev)
        with torch.no_grad():
            feats, sem, off, emb, pred = model.backbone_and_heads()
        assert feats.shape == (n, 16) and sem.shape == (n, 9) and bool(torch.isfinite(sem).all())
        np.testing.assert_allclose(torch.exp(sem).sum(1).cpu().numpy(), 1.0, rtol=1e-5)


def test_coordinates_at_the_ends_of_the_key_range():
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    dev = torch.device("cuda")
    c = torch.tensor([[0, 32767, 32767, 32767], [0, 32766, 32767, 32767], [0, -32768, -32768, -32768],
                      [1, -32768, -32768, -32768], [0, 0, 0, 0]], dtype=torch.int32)
    cm = ME.CoordinateManager(c.to(dev))
    nbr = cm.kernel_map_rows(1, 1, 3, 1).cpu().numpy()
    assert (nbr >= 0).sum() == 5 + 2                              # every row finds itself; one adjacent pair, both ways
    x = ME.SparseTensor(torch.randn(5, 4), coordinates=c, device=dev)
    assert torch.equal(x.C.cpu(), c)                              # caller order preserved
    with pytest.raises(Exception):
        ME.SparseTensor(torch.randn(1, 4), coordinates=torch.tensor([[0, 40000, 0, 0]], dtype=torch.int32), device=dev)
    with pytest.raises(ValueError):
        ME.SparseTensor(torch.randn(2, 4), coordinates=torch.tensor([[0, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32), device=dev)


def test_weight_gradient_pair_lists_degenerate_maps():
    """pair lists of maps with no rows, no pairs at all, one pair, 8 offsets (2x2x2 kernels) and a single row: the lists hold
    exactly the map's pairs and dW equals the definition; a training step on a 2-voxel input runs through the fused conv + BN
    node and the pair-major weight gradient (two batch elements of one voxel); a 1-voxel input raises torch's "more than 1 value per channel" ValueError, as
    BatchNorm1d under ME.MinkowskiBatchNorm does in the reference."""
    from panopticsegforlargescalepointcloud_amd import ops
    from panopticsegforlargescalepointcloud_amd.applications import Data
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for K, rows, fill in [(27, 0, None), (27, 5, "none"), (27, 1300, "one"), (8, 777, "rand"), (27, 1, "self"), (1, 40, "self")]:
        nbr = torch.full((K, rows), -1, dtype=torch.int32, device=dev)
        n_in = max(rows, 1) + 3
        if fill == "one":
            nbr[K // 2, rows - 1] = 2
        elif fill == "rand":
            nbr = torch.where(torch.rand((K, rows), device=dev, generator=g) < 0.3,
                              torch.randint(0, n_in, (K, rows), device=dev, generator=g, dtype=torch.int32), nbr)
        elif fill == "self":
            nbr[K // 2] = torch.arange(rows, device=dev, dtype=torch.int32)
        x = torch.randn(n_in, 8, device=dev, generator=g)
        dy = torch.randn(rows, 12, device=dev, generator=g)
        wp = ops.wgrad_pairs(nbr, K)
        assert int(wp.tile_start[-1]) == int((nbr >= 0).sum())
        dw = ops.spconv_bwd_weight_pairs(x, dy, wp)
        want = torch.zeros(K, 8, 12, dtype=torch.float64, device=dev)
        for k in range(K):
            r = torch.nonzero(nbr[k] >= 0).view(-1)
            want[k] = x[nbr[k][r].long()].double().t() @ dy[r].double()
        np.testing.assert_allclose(dw.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    model, cfg, DS = _model()
    model.train()
    for n in (2, 1):   # two batch elements of one voxel each: every level keeps two rows
        coords = torch.zeros(n, 3, dtype=torch.int32)
        data = Data(pos=coords.float() * 0.05, coords=coords, x=torch.randn(n, 4), batch=torch.arange(n),
                    y=torch.zeros(n, dtype=torch.long), instance_labels=torch.ones(n, dtype=torch.long),
                    instance_mask=torch.ones(n, dtype=torch.bool), vote_label=torch.zeros(n, 3),
                    center_label=torch.zeros(n, 3), num_instances=torch.tensor([1]))
        model.set_input(data, dev)
        if n == 1:
            with pytest.raises(ValueError, match="more than 1 value per channel"):
                model.forward(epoch=1)
            continue
        model.forward(epoch=1)
        model.backward(1)
        assert np.isfinite(float(model.loss.detach()))
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
