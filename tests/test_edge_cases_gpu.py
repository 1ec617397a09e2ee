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
