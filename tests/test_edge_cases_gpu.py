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


def test_maps_built_under_no_grad_serve_a_later_training_step(monkeypatch):
    """A coordinate manager first used under no_grad caches the transposed stride-2 map in its 8-wide inference form ([8, n]);
    autograd must never see it (the weight gradient and the pair lists index 27 * n entries): the same manager then serves a
    training step through the dense twin, with the gradients of a manager that was never used under no_grad."""
    import bruteforce as bf
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, modules as M
    monkeypatch.setattr(ME, "MAP_ORDER_MIN_ROWS", 500)
    rng = np.random.default_rng(3)
    coords = torch.from_numpy(bf.surface_coords(rng, n_batch=2, n=3000, extent=40)).cuda()
    x0 = torch.randn(len(coords), 16, device="cuda")
    torch.manual_seed(2)
    down = M.ResNetDown(down_conv_nn=[16, 32], kernel_size=3, stride=2, N=1).cuda()
    up = M.ResNetUp(up_conv_nn=[32 + 32, 16], kernel_size=3, stride=2, N=1).cuda()
    bns = [m for m in list(down.modules()) + list(up.modules()) if isinstance(m, torch.nn.BatchNorm1d)]

    def step(cm=None):
        for b_ in bns:
            b_.reset_running_stats()
        down.train(), up.train()
        down.zero_grad(), up.zero_grad()
        if cm is None:
            x = x0.clone().requires_grad_(True)
            st = ME.SparseTensor(features=x, coordinates=coords, device="cuda")
        else:  # features handed over in the manager's internal row order
            x = cm.to_internal(x0).clone().requires_grad_(True)
            st = ME.SparseTensor(x, coordinate_manager=cm, tensor_stride=1)
        h = down(st)
        y = up(h, h)
        (y.F ** 2).mean().backward()
        dx = x.grad if cm is None else cm.to_caller(x.grad)
        return y.F.detach().clone(), dx.clone(), {n: p.grad.clone() for n, p in list(down.named_parameters()) + list(up.named_parameters())}

    down.eval(), up.eval()
    with torch.no_grad():  # inference pre-pass: the manager caches the 8-wide map
        st = ME.SparseTensor(features=x0, coordinates=coords, device="cuda")
        h = down(st)
        up(h, h)
    cm = st.coordinate_manager
    assert any(getattr(m, "pp_t8", False) for m in cm.maps.values()), "the pre-pass built no 8-wide map: the test is void"
    y_ref, dx_ref, g_ref = step()        # a fresh manager that never ran under no_grad
    y, dx, g = step(cm)                  # the manager of the pre-pass
    assert any(hasattr(m, "pp_dense") for m in cm.maps.values())
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), dx_ref.cpu().numpy(), rtol=1e-4, atol=1e-7)
    for n in g_ref:
        np.testing.assert_allclose(g[n].cpu().numpy(), g_ref[n].cpu().numpy(), rtol=2e-4, atol=1e-6, err_msg=n)


def test_packed_weight_cache_follows_a_storage_swap():
    """`p.data = ...` (EMA / SWA swaps, vector_to_parameters, to_empty) replaces a parameter's storage without touching its
    version counter or the optimizer epoch: the batched packed-weight cache keys on the data pointer as well."""
    from panopticsegforlargescalepointcloud_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(27, 16, 32, device=dev))
    q = torch.nn.Parameter(torch.randn(27, 32, 16, device=dev))
    cache = ops._PackedWeights()
    a0 = cache.get(p, False, False).clone()
    cache.get(q, False, False)
    assert torch.equal(a0, ops.pack_weight(p))
    new = torch.randn(27, 16, 32, device=dev)
    new0 = new.clone()                # (p shares `new`'s storage from here on)
    v = p._version
    p.data = new                      # same version, new storage
    assert p._version == v
    a1 = cache.get(p, False, False)
    assert torch.equal(a1, ops.pack_weight(new0)) and not torch.equal(a1, a0)
    assert torch.equal(cache.get(q, False, False), ops.pack_weight(q))      # the other entry was re-packed from ITS storage
    with torch.no_grad():
        p.mul_(2.0)                   # in-place: version bump, same storage
    assert torch.equal(cache.get(p, False, False), ops.pack_weight(new0 * 2.0))


def test_resblock_with_a_frozen_second_batchnorm_counts_the_first_once():
    """Training-mode ResBlock whose second BatchNorm is in eval mode (partially frozen block): the fused first pair has
    already updated its running statistics when the second pair turns out not to be the plain case -- the block continues
    from there module by module instead of running (and counting) the first pair again."""
    import bruteforce as bf
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, modules as M
    rng = np.random.default_rng(5)
    coords = torch.from_numpy(bf.surface_coords(rng, n_batch=1, n=1500, extent=30)).cuda()
    x0 = torch.randn(len(coords), 16, device="cuda")
    torch.manual_seed(1)
    blk = M.ResBlock(16, 32, ME.MinkowskiConvolution).cuda().train()
    blk.block[4].eval()               # freeze the second BatchNorm only
    ref = M.ResBlock(16, 32, ME.MinkowskiConvolution).cuda().train()
    ref.load_state_dict(blk.state_dict())
    ref.block[4].eval()
    st = ME.SparseTensor(features=x0, coordinates=coords, device="cuda")
    y = blk(st)
    assert int(blk.block[1].bn.num_batches_tracked) == 1 and int(blk.block[4].bn.num_batches_tracked) == 0
    # module by module (what the fallback of the whole block does), on a copy
    h = ref.block(st)
    y_ref = h + ref.downsample(st)
    np.testing.assert_allclose(y.F.detach().cpu().numpy(), y_ref.F.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(blk.block[1].bn.running_mean, ref.block[1].bn.running_mean, rtol=1e-6, atol=1e-7)
