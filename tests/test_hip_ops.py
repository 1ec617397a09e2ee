"""GPU parity tests: every C-ABI entry point against the CPU oracle on the same seeded inputs
(bit-exact for integer/index work, 1e-4 for float32), plus the golden fixtures."""
import os

import numpy as np
import pytest
import torch

import bruteforce as bf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from panopticsegforlargescalepointcloud_amd import ops as o
    return o


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def surface(rng, n=4000, n_batch=3, extent=80):
    return bf.surface_coords(rng, n_batch=n_batch, n=n, extent=extent)


# ------------------------------------------------------------------ coordinates
def test_hash_build_counts_duplicates(ops, oracle):
    rng = np.random.default_rng(0)
    coords = bf.surface_coords(rng, n=3000, extent=60, dup=True)
    _, ndup = oracle.hash_first_rows(coords)
    table, got = ops.hash_build(dev(coords))
    assert got == ndup
    uniq = bf.surface_coords(rng, n=3000, extent=60)
    assert ops.hash_build(dev(uniq))[1] == 0


@pytest.mark.parametrize("ts", [2, 4, 8])
def test_stride_coords_bit_exact(ops, oracle, ts):
    rng = np.random.default_rng(1)
    coords = surface(rng)
    coords[:, 1:] *= ts // 2
    want, want_f2c = oracle.stride_coords(coords, ts)
    out, table, f2c = ops.stride_coords(dev(coords), ts)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(f2c.cpu().numpy(), want_f2c)


def test_kernel_maps_bit_exact(ops, oracle):
    rng = np.random.default_rng(2)
    fine = surface(rng)
    dfine = dev(fine)
    table, _ = ops.hash_build(dfine)
    for ksize, sign in [(3, 1), (3, -1), (1, 1)]:
        got = ops.kernel_map(dfine, table, ksize, 1, sign).cpu().numpy()
        assert np.array_equal(got, oracle.kernel_map(fine, fine, ksize, 1, sign))
    coarse, ctable, _ = ops.stride_coords(dfine, 2)
    c_np = coarse.cpu().numpy()
    down = ops.kernel_map(coarse, table, 3, 1, 1).cpu().numpy()
    up = ops.kernel_map(dfine, ctable, 3, 1, -1).cpu().numpy()
    assert np.array_equal(down, oracle.kernel_map(c_np, fine, 3, 1, 1))
    assert np.array_equal(up, oracle.kernel_map(fine, c_np, 3, 1, -1))
    # second level (tensor stride 2 -> 4), step = 2
    coarse2, _, _ = ops.stride_coords(coarse, 4)
    got = ops.kernel_map(coarse2, ctable, 3, 2, 1).cpu().numpy()
    assert np.array_equal(got, oracle.kernel_map(coarse2.cpu().numpy(), c_np, 3, 2, 1))


def test_empty_inputs(ops):
    e = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    table, nd = ops.hash_build(e)
    assert nd == 0
    out, t2, f2c = ops.stride_coords(e, 2)
    assert out.shape[0] == 0
    nbr = ops.kernel_map(e, table, 3, 1, 1)
    assert nbr.shape == (27, 0)


# ------------------------------------------------------------------ convolution
CONV_CASES = [(4, 16, 3), (16, 16, 3), (16, 32, 3), (32, 48, 3), (48, 48, 3), (80, 96, 3), (112, 112, 3), (192, 80, 3),
              (16, 32, 1), (192, 80, 1), (8, 24, 3), (12, 20, 1)]


@pytest.mark.parametrize("cin,cout,ksize", CONV_CASES)
def test_spconv_fwd_matches_oracle(ops, oracle, cin, cout, ksize):
    rng = np.random.default_rng(3)
    coords = surface(rng, n=1500, n_batch=2, extent=40)
    n = len(coords)
    nbr = oracle.kernel_map(coords, coords, ksize, 1, 1)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    K = ksize ** 3
    W = (rng.normal(size=(K, cin, cout)) / np.sqrt(K * cin / 4)).astype(np.float32)
    want = oracle.spconv_fwd(x, W, nbr, n)
    packed = ops.pack_weight(dev(W))
    got = ops.spconv_fwd(dev(x), packed, dev(nbr) if ksize == 3 else None, n, cout, K)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_spconv_fwd_cat_epilogue_and_ragged_tail(ops, oracle):
    rng = np.random.default_rng(4)
    for n_pts in [1, 17, 130, 1111]:
        coords = surface(rng, n=n_pts, n_batch=1, extent=30)
        n = len(coords)
        nbr = oracle.kernel_map(coords, coords, 3, 1, 1)
        a = rng.normal(size=(n, 32)).astype(np.float32)
        b = rng.normal(size=(n, 32)).astype(np.float32)
        W = (rng.normal(size=(27, 64, 16)) * 0.1).astype(np.float32)
        sc = rng.normal(size=16).astype(np.float32)
        sh = rng.normal(size=16).astype(np.float32)
        res = rng.normal(size=(n, 16)).astype(np.float32)
        want = oracle.spconv_fwd(a, W, nbr, n, in1=b, scale=sc, shift=sh, relu=True, residual=res)
        got = ops.spconv_fwd(dev(a), ops.pack_weight(dev(W)), dev(nbr), n, 16, 27, in1=dev(b), scale=dev(sc),
                             shift=dev(sh), relu=True, residual=dev(res))
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_spconv_strided_and_transposed(ops, oracle):
    rng = np.random.default_rng(5)
    fine = surface(rng, n=2000, n_batch=2, extent=40)
    coarse, _ = oracle.stride_coords(fine, 2)
    down = oracle.kernel_map(coarse, fine, 3, 1, 1)
    up = oracle.kernel_map(fine, coarse, 3, 1, -1)
    x = rng.normal(size=(len(fine), 16)).astype(np.float32)
    W = (rng.normal(size=(27, 16, 16)) * 0.1).astype(np.float32)
    y_want = oracle.spconv_fwd(x, W, down, len(coarse))
    y = ops.spconv_fwd(dev(x), ops.pack_weight(dev(W)), dev(down), len(coarse), 16, 27)
    np.testing.assert_allclose(y.cpu().numpy(), y_want, rtol=1e-4, atol=1e-4)
    W2 = (rng.normal(size=(27, 16, 32)) * 0.1).astype(np.float32)
    z_want = oracle.spconv_fwd(y_want, W2, up, len(fine))
    z = ops.spconv_fwd(dev(y_want), ops.pack_weight(dev(W2)), dev(up), len(fine), 32, 27)
    np.testing.assert_allclose(z.cpu().numpy(), z_want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cin,cout", [(64, 16), (64, 32), (32, 16), (48, 96), (16, 128), (16, 192), (20, 40), (96, 96),
                                      (48, 16), (48, 32), (32, 48), (32, 64), (16, 80), (16, 112), (16, 144), (16, 160),
                                      (16, 176), (8, 8), (4, 4), (4, 16), (224, 96), (6, 10), (18, 7)])
def test_spconv_weight_gradient_shapes(ops, oracle, cin, cout):
    """every (offsets per wave, ci tiles per wave, co tiles) instantiation of the LDS-staged weight gradient (widths that
    are multiples of 4) and the per-operand-load kernel behind it (the other widths), ragged row counts, a strided map
    (n_in != n_out, many missing neighbours) and ragged channel counts; float32 accumulation order differs from the
    oracle's, hence the tolerance."""
    rng = np.random.default_rng(60 + cin + cout)
    fine = surface(rng, n=1500, n_batch=2, extent=36)
    coarse, _ = oracle.stride_coords(fine, 2)
    for out_c, in_c, sign, st in [(fine, fine, 1, 1), (coarse, fine, 1, 1), (fine, coarse, -1, 1)]:
        nbr = oracle.kernel_map(out_c, in_c, 3, st, sign)
        x = rng.normal(size=(len(in_c), cin)).astype(np.float32)
        g = rng.normal(size=(len(out_c), cout)).astype(np.float32)
        want = np.zeros((27, cin, cout))
        for k in range(27):
            m = nbr[k] >= 0
            want[k] = x[nbr[k][m]].astype(np.float64).T @ g[m].astype(np.float64)
        dw = ops.spconv_bwd_weight(dev(x), dev(g), dev(nbr), 27)
        np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=2e-4, atol=2e-3)
        # pair-major form: the per-offset lists are exactly the pairs of the map in row order ...
        wp = ops.wgrad_pairs(dev(nbr), 27)
        ts, pairs, T = wp.tile_start.cpu().numpy(), wp.pairs.cpu().numpy(), (nbr.shape[1] + 1023) // 1024
        assert ts[0] == 0 and ts[-1] == (nbr >= 0).sum() and len(ts) == 27 * T + 1
        for k in range(27):
            rows = np.nonzero(nbr[k] >= 0)[0]
            np.testing.assert_array_equal(pairs[ts[k * T]:ts[(k + 1) * T]], np.stack([rows, nbr[k][rows]], 1))
        np.testing.assert_allclose(ops.spconv_bwd_weight_pairs(dev(x), dev(g), wp).cpu().numpy(), want, rtol=2e-4, atol=2e-3)
        # ... and a slot-ordered map (rows permuted, row_order = slot -> output row) gives the same dW from the same dout
        perm = rng.permutation(nbr.shape[1]).astype(np.int32)
        wp = ops.wgrad_pairs(dev(np.ascontiguousarray(nbr[:, perm])), 27, row_order=dev(perm))
        np.testing.assert_allclose(ops.spconv_bwd_weight_pairs(dev(x), dev(g), wp).cpu().numpy(), want, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (96, 96), (20, 40)])
def test_weight_gradient_without_atomics_is_bit_reproducible(ops, oracle, monkeypatch, cin, cout):
    """pp_spconv_bwd_weight_pairs_det (block partials in a workspace + ordered reduction, the default of the training path) gives
    the same BITS run after run and agrees with the float-atomic form and the float64 reference; offsets without pairs and
    blocks past the end of an offset's list contribute nothing."""
    rng = np.random.default_rng(7 + cin)
    fine = surface(rng, n=9000, n_batch=2, extent=60)
    coarse, _ = oracle.stride_coords(fine, 2)
    for out_c, in_c, sign in [(fine, fine, 1), (coarse, fine, 1)]:
        nbr = oracle.kernel_map(out_c, in_c, 3, 1, sign)
        nbr[5] = -1                                             # an offset without a single pair
        x = rng.normal(size=(len(in_c), cin)).astype(np.float32)
        g = rng.normal(size=(len(out_c), cout)).astype(np.float32)
        want = np.zeros((27, cin, cout))
        for k in range(27):
            m = nbr[k] >= 0
            want[k] = x[nbr[k][m]].astype(np.float64).T @ g[m].astype(np.float64)
        wp = ops.wgrad_pairs(dev(nbr), 27)
        assert ops.WGRAD_DETERMINISTIC is True
        runs = [ops.spconv_bwd_weight_pairs(dev(x), dev(g), wp) for _ in range(4)]
        for r in runs[1:]:
            assert torch.equal(r, runs[0])
        np.testing.assert_allclose(runs[0].cpu().numpy(), want, rtol=2e-4, atol=2e-3)
        assert float(runs[0][5].abs().max()) == 0.0
        monkeypatch.setattr(ops, "WGRAD_DETERMINISTIC", False)
        atomic = ops.spconv_bwd_weight_pairs(dev(x), dev(g), wp)
        monkeypatch.setattr(ops, "WGRAD_DETERMINISTIC", True)
        np.testing.assert_allclose(atomic.cpu().numpy(), runs[0].cpu().numpy(), rtol=1e-4, atol=1e-3)
        bf = [ops.spconv_bwd_weight_pairs(dev(x), dev(g), wp, bf16=True) for _ in range(2)]
        assert torch.equal(bf[0], bf[1])


@pytest.mark.parametrize("cin,cout,n", [(96, 96, 900), (112, 112, 300), (64, 80, 4000), (16, 16, 40), (160, 64, 2500)])
def test_spconv_split_k_small_launches(ops, oracle, cin, cout, n):
    """small launches take the split-K path (offsets spread over several waves, partials added in a fixed order by
    k_spconv_split_reduce with the fused epilogue): same result as the oracle, bit-identical run to run, and the
    registered scratch being too small falls back to the unsplit kernel."""
    rng = np.random.default_rng(90 + cin + n)
    coords = surface(rng, n=n, n_batch=2, extent=24)
    n = len(coords)
    nbr = oracle.kernel_map(coords, coords, 3, 1, 1)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(n, cout)).astype(np.float32)
    want = oracle.spconv_fwd(x, W, nbr, n, scale=sc, shift=sh, relu=True, residual=res)
    run = lambda: ops.spconv_fwd(dev(x), ops.pack_weight(dev(W)), dev(nbr), n, cout, 27, scale=dev(sc), shift=dev(sh),
                                 relu=True, residual=dev(res))
    got = run()
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert torch.equal(got, run())
    lib = ops._lib.load()
    try:
        assert lib.pp_spconv_set_scratch(None, 0) == 0
        unsplit = run()
    finally:
        ops._CONV_SCRATCH["key"] = None  # re-register on the next call
    np.testing.assert_allclose(unsplit.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    if cin * 27 > 500:
        assert not torch.equal(unsplit, got)  # a different summation order: the split path really ran


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 48), (64, 64), (96, 112), (48, 16)])
def test_spconv_bf16_entries_match_oracle_on_rounded_operands(ops, oracle, cin, cout):
    """pp_spconv_fwd_bf16 / pp_spconv_bwd_weight_bf16: operands rounded to bfloat16, fp32 accumulation.  Products of
    bfloat16 numbers are exact in fp32, so against the fp32 oracle fed with pre-rounded operands only the summation order
    differs (tolerance 1e-4 relative to the output scale); against the un-rounded fp32 result the error is the bf16
    rounding (~2^-8 per operand), checked loosely.  Includes the fused epilogue, a second source and a strided map."""
    rng = np.random.default_rng(70 + cin + cout)
    fine = surface(rng, n=1600, n_batch=2, extent=36)
    coarse, _ = oracle.stride_coords(fine, 2)
    for out_c, in_c, sign in [(fine, fine, 1), (coarse, fine, 1), (fine, coarse, -1)]:
        nbr = oracle.kernel_map(out_c, in_c, 3, 1, sign)
        n_in, n_out = len(in_c), len(out_c)
        x = rng.normal(size=(n_in, cin)).astype(np.float32)
        W = (rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.normal(size=cout).astype(np.float32)
        res = rng.normal(size=(n_out, cout)).astype(np.float32)
        want = oracle.spconv_fwd(oracle.round_bf16(x), oracle.round_bf16(W), nbr, n_out, scale=sc, shift=sh, relu=True,
                                 residual=res)
        got = ops.spconv_fwd(dev(x), ops.pack_weight(dev(W)), dev(nbr), n_out, cout, 27, scale=dev(sc), shift=dev(sh),
                             relu=True, residual=dev(res), bf16=True).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
        full = oracle.spconv_fwd(x, W, nbr, n_out, scale=sc, shift=sh, relu=True, residual=res)
        assert np.abs(got - full).max() < 0.05 * max(1.0, float(np.abs(full).max()))
        assert np.abs(got - full).max() > 0  # really a different arithmetic
        g = rng.normal(size=(n_out, cout)).astype(np.float32)
        xr, gr = oracle.round_bf16(x).astype(np.float64), oracle.round_bf16(g).astype(np.float64)
        want_dw = np.zeros((27, cin, cout))
        for k in range(27):
            m = nbr[k] >= 0
            want_dw[k] = xr[nbr[k][m]].T @ gr[m]
        dw = ops.spconv_bwd_weight(dev(x), dev(g), dev(nbr), 27, bf16=True).cpu().numpy()
        np.testing.assert_allclose(dw, want_dw, rtol=2e-4, atol=2e-3)
        dw = ops.spconv_bwd_weight_pairs(dev(x), dev(g), ops.wgrad_pairs(dev(nbr), 27), bf16=True).cpu().numpy()
        np.testing.assert_allclose(dw, want_dw, rtol=2e-4, atol=2e-3)
    if cin % 32 == 0:  # two sources (ME.cat fused)
        nbr = oracle.kernel_map(fine, fine, 3, 1, 1)
        x0 = rng.normal(size=(len(fine), cin // 2)).astype(np.float32)
        x1 = rng.normal(size=(len(fine), cin // 2)).astype(np.float32)
        want = oracle.spconv_fwd(oracle.round_bf16(x0), oracle.round_bf16(W), nbr, len(fine), in1=oracle.round_bf16(x1))
        got = ops.spconv_fwd(dev(x0), ops.pack_weight(dev(W)), dev(nbr), len(fine), cout, 27, in1=dev(x1), bf16=True)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * float(np.abs(want).max()))


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 48), (96, 32), (4, 16)])
def test_spconv_backward_matches_oracle(ops, oracle, cin, cout):
    rng = np.random.default_rng(6)
    coords = surface(rng, n=1800, n_batch=2, extent=40)
    n = len(coords)
    nbr = oracle.kernel_map(coords, coords, 3, 1, 1)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    g = rng.normal(size=(n, cout)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32)
    din_want, dw_want = oracle.spconv_bwd(x, g, W, nbr)
    dw = ops.spconv_bwd_weight(dev(x), dev(g), dev(nbr), 27)
    np.testing.assert_allclose(dw.cpu().numpy(), dw_want, rtol=2e-4, atol=2e-3)
    # input gradient = forward conv of dout with W^T over the mirrored map
    packedT = ops.pack_weight(dev(W), transpose=True)
    din = ops.spconv_fwd(dev(g), packedT, dev(nbr[::-1].copy()), n, cin, 27)
    np.testing.assert_allclose(din.cpu().numpy(), din_want, rtol=1e-4, atol=1e-4)


def test_bn_train_fwd_bwd_matches_torch_float64(ops):
    """pp_bn_train_fwd / pp_bn_train_bwd against torch's BatchNorm (float64, CPU autograd) incl. the fused ReLU and the
    running statistics; two runs are bit-identical (no atomics).  Tolerance: float32 rounding of the outputs."""
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    for n, c in [(2, 4), (37, 6), (5000, 32), (4099, 96), (100_003, 64), (3001, 256), (777, 3)]:
        for relu in (False, True):
            x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c) * 2).astype(np.float32)
            w = rng.uniform(0.5, 1.5, c).astype(np.float32)
            b = rng.normal(size=c).astype(np.float32)
            dy = rng.normal(size=(n, c)).astype(np.float32)
            rm0 = rng.normal(size=c).astype(np.float32)
            rv0 = rng.uniform(0.5, 2, c).astype(np.float32)
            xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
            wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
            bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
            rm, rv = torch.tensor(rm0, dtype=torch.float64), torch.tensor(rv0, dtype=torch.float64)
            yt = F.batch_norm(xt, rm, rv, wt, bt, training=True, momentum=0.1, eps=1e-5)
            if relu:
                yt = torch.relu(yt)
            yt.backward(torch.tensor(dy, dtype=torch.float64))
            grm, grv = dev(rm0), dev(rv0)
            nbt = torch.tensor(41, dtype=torch.int64, device=grm.device)   # nn.BatchNorm's num_batches_tracked
            y, mean, rstd = ops.bn_train_fwd(dev(x), dev(w), dev(b), 1e-5, 0.1, grm, grv, relu, nbt)
            assert int(nbt) == 42
            np.testing.assert_allclose(y.cpu().numpy(), yt.detach().numpy(), rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(grm.cpu().numpy(), rm.numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(grv.cpu().numpy(), rv.numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(mean.cpu().numpy(), x.astype(np.float64).mean(0), rtol=1e-9, atol=1e-9)
            dx, dw, db = ops.bn_train_bwd(dev(x), dev(dy), y if relu else None, dev(w), mean, rstd)
            scale = max(1.0, float(np.abs(xt.grad.numpy()).max()))
            np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=2e-5 * scale)
            np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=1e-3)
            np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-3)
            y2, mean2, rstd2 = ops.bn_train_fwd(dev(x), dev(w), dev(b), 1e-5, 0.1, None, None, relu)
            assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
            dx2, dw2, db2 = ops.bn_train_bwd(dev(x), dev(dy), y if relu else None, dev(w), mean, rstd)
            assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    # affine=False and the argument checks
    x = dev(rng.normal(size=(100, 8)).astype(np.float32))
    y, mean, rstd = ops.bn_train_fwd(x, None, None, 1e-5, 0.1, None, None, False)
    np.testing.assert_allclose(y.cpu().numpy().mean(0), 0, atol=1e-6)
    np.testing.assert_allclose(y.cpu().numpy().std(0), 1, atol=1e-3)
    with pytest.raises(Exception):
        ops.bn_train_fwd(x[:0], None, None, 1e-5, 0.1, None, None, False)


def test_bn_pieces_and_heads(ops, oracle):
    rng = np.random.default_rng(7)
    for c in [16, 48, 96, 192]:
        x = rng.normal(size=(5000, c)).astype(np.float32)
        s, ss = ops.channel_stats(dev(x))
        ws, wss = oracle.channel_stats(x)
        np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(ss.cpu().numpy(), wss, rtol=1e-10, atol=1e-8)
        dy = rng.normal(size=(5000, c)).astype(np.float32)
        a, b = ops.bn_bwd_reduce(dev(x), dev(dy))
        np.testing.assert_allclose(a.cpu().numpy(), dy.astype(np.float64).sum(0), rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(b.cpu().numpy(), (dy.astype(np.float64) * x).sum(0), rtol=1e-9, atol=1e-8)
        sc = rng.normal(size=c).astype(np.float32)
        sh = rng.normal(size=c).astype(np.float32)
        r = rng.normal(size=(5000, c)).astype(np.float32)
        for act in [0, 1, 2]:
            got = ops.affine_act(dev(x), dev(sc), dev(sh), act=act, slope=0.2, residual=dev(r))
            want = oracle.affine_act(x, sc, sh, act=act, slope=0.2, residual=r)
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    x = rng.normal(size=(3000, 16)).astype(np.float32)
    w1 = rng.normal(size=(16, 16)).astype(np.float32) * 0.3
    sc = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    sh = rng.normal(size=16).astype(np.float32)
    for cout, ls in [(9, True), (3, False), (5, False), (1, False)]:
        w2 = rng.normal(size=(cout, 16)).astype(np.float32) * 0.3
        b2 = rng.normal(size=cout).astype(np.float32)
        want, wam = oracle.head_mlp(x, w1, sc, sh, w2, b2, log_softmax=ls, want_argmax=True)
        got, am = ops.head_mlp(dev(x), dev(w1), dev(sc), dev(sh), dev(w2), dev(b2), log_softmax=ls, want_argmax=True)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
        assert (am.cpu().numpy() != wam).mean() < 1e-3  # argmax may flip only on float ties
    # all heads in one pass (pp_heads), with and without a row index: bit-identical to the one-head launches
    hs = []
    for cout, ls, am in [(9, True, True), (3, False, False), (5, False, False)]:
        hs.append((dev(w1 * (1 + cout)), dev(sc), dev(sh), dev(rng.normal(size=(cout, 16)).astype(np.float32) * 0.3),
                   dev(rng.normal(size=cout).astype(np.float32)) if cout != 5 else None, ls, am))
    index = dev(rng.permutation(3000)[:2500].astype(np.int64))
    for idx in [None, index]:
        xs = dev(x) if idx is None else dev(x)[idx]
        for k in (1, 2, 3):
            res = ops.heads(dev(x), hs[:k], index=idx)
            for (y, a), h in zip(res, hs[:k]):
                wy, wa = ops.head_mlp(xs.contiguous(), *h[:5], log_softmax=h[5], want_argmax=True)
                assert torch.equal(y, wy)
                assert (a is None) == (not h[6]) and (a is None or torch.equal(a, wa))
    ops.gather_rows_check()
    ops.heads(dev(x), hs[:1], index=dev(np.array([0, 3000], np.int64)))
    with pytest.raises(Exception, match="out of range"):
        ops.gather_rows_check()
    assert ops.heads(dev(x[:0]), hs[:2])[1][0].shape == (0, 3)


# ------------------------------------------------------------------ region growing
def _blobs3(rng, n_blobs, pts, spread, sigma):
    cen = rng.uniform(-spread, spread, size=(n_blobs, 3))
    ids = rng.integers(0, n_blobs, size=pts)
    return (cen[ids] + rng.normal(0, sigma, size=(pts, 3))).astype(np.float32), ids


@pytest.mark.parametrize("nsample,sigma", [(200, 0.10), (16, 0.10), (4, 0.10), (200, 0.02), (32, 0.02)])
def test_region_grow_bit_exact(ops, oracle, nsample, sigma):
    rng = np.random.default_rng(8)
    n = 20000
    pos, _ = _blobs3(rng, 60, n, 6.0, sigma)
    labels = rng.integers(0, 5, size=n)
    batch = np.sort(rng.integers(0, 3, size=n))
    ignore = [0, 3]
    want, want_pc = oracle.region_grow(pos, labels, batch, ignore, nsample=nsample, radius=0.15, min_cluster_size=10)
    csr, pc = ops.region_grow_csr(dev(pos), dev(labels), dev(batch), torch.tensor(ignore), nsample, 0.15, 10, 5)
    got = [c.cpu().numpy() for c in csr.to_list()]
    assert len(got) == len(want) and len(want) > 0
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert np.array_equal(pc.cpu().numpy(), want_pc)


def test_region_grow_pileup_overflows_lds_buffer(ops, oracle):
    """dense pile-ups: 2600 and 5000 candidates per cell go to the large-buffer instance of the cell kernel (1536 < n <= 8192),
    9000 to the per-query kernel's global re-scan selection (> 4096 hits per query; its LDS bisection is exercised by the
    aliased cells below)."""
    rng = np.random.default_rng(9)
    n = 16600
    pos = rng.normal(0, 0.01, size=(n, 3)).astype(np.float32)
    pos[5000:7600] += 5.0
    pos[7600:] += 10.0
    labels = np.ones(n, np.int64)
    batch = np.zeros(n, np.int64)
    want, _ = oracle.region_grow(pos, labels, batch, [], nsample=200, radius=0.2, min_cluster_size=10)
    csr, _ = ops.region_grow_csr(dev(pos), dev(labels), dev(batch), torch.zeros(0, dtype=torch.int64), 200, 0.2, 10, 2)
    got = [c.cpu().numpy() for c in csr.to_list()]
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_region_grow_aliased_cells(ops, oracle):
    """cell keys wrap at 4096 cells per axis and fold (batch, class) above 2^27: blobs exactly 4096 cells apart, and batch
    ids beyond 2^19, land behind the same key -- those cells must go through the per-query path with exact results."""
    rng = np.random.default_rng(12)
    radius = 0.1
    base = rng.normal(0, 0.08, size=(1500, 3)).astype(np.float32)
    far = base[:700] + np.float32(4096 * radius) * np.array([1, 0, 0], np.float32)   # same wrapped cell coordinates
    far2 = base[:500] + np.float32(4096 * radius) * np.array([0, -1, 1], np.float32)
    pos = np.concatenate([base, far, far2, base[:600] + 0.01, base[:400] - 0.01])
    labels = np.ones(len(pos), np.int64)
    batch = np.concatenate([np.zeros(1500 + 700 + 500, np.int64), np.full(600, 524288 + 3, np.int64), np.full(400, 2 * 524288 + 3, np.int64)])
    for nsample in (16, 200):
        want, want_pc = oracle.region_grow(pos, labels, batch, [], nsample=nsample, radius=radius, min_cluster_size=10)
        csr, pc = ops.region_grow_csr(dev(pos), dev(labels), dev(batch), torch.zeros(0, dtype=torch.int64), nsample, radius, 10, 2)
        got = [c.cpu().numpy() for c in csr.to_list()]
        assert len(got) == len(want) and len(want) >= 3
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        assert np.array_equal(pc.cpu().numpy(), want_pc)


def test_region_grow_degenerate(ops):
    e = torch.zeros((0, 3), device="cuda")
    l = torch.zeros(0, dtype=torch.int64, device="cuda")
    csr, pc = ops.region_grow_csr(e, l, l, torch.zeros(0, dtype=torch.int64), 16, 0.1, 10, 2)
    assert csr.n == 0 and csr.to_list() == []
    # everything ignored
    pos = torch.rand(100, 3, device="cuda")
    lab = torch.zeros(100, dtype=torch.int64, device="cuda")
    csr, pc = ops.region_grow_csr(pos, lab, lab, torch.tensor([0]), 16, 0.1, 10, 2)
    assert csr.n == 0 and bool((pc == -1).all())


# ------------------------------------------------------------------ mean shift
def test_meanshift_matches_reference_goldens(ops):
    z = np.load(os.path.join(GOLD, "meanshift_cases.npz"))
    for name in z["names"].tolist():
        x = z["x_" + name]
        labels, ncl, _ = ops.meanshift(dev(x), [0, len(x)], float(z["bw_" + name]))
        want = z["labels_" + name]
        assert int(ncl[0]) == want.max() + 1, name
        assert np.array_equal(bf.canon_partition(labels.cpu().numpy()), bf.canon_partition(want)), name


def test_meanshift_batched_matches_oracle(ops, oracle):
    rng = np.random.default_rng(10)
    xs, offs = [], [0]
    for s, (n, k) in enumerate([(3000, 20), (2, 1), (5000, 40), (0, 1), (1500, 5)]):
        cen = rng.normal(0, 3.0, size=(k, 5))
        x = cen[rng.integers(0, k, size=n)] + rng.normal(0, 0.15, size=(n, 5))
        xs.append(x.astype(np.float32))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    wl, wn, wc = oracle.meanshift(x, offs, 0.6)
    labels, ncl, centers = ops.meanshift(dev(x), offs, 0.6, want_centers=True)
    assert np.array_equal(ncl.cpu().numpy(), wn)
    assert np.array_equal(labels.cpu().numpy(), wl)
    np.testing.assert_allclose(centers.cpu().numpy(), wc, atol=2e-3)


def test_meanshift_many_seeds_per_sample(ops, oracle):
    """> 1024 bin seeds in a sample take the radix-sort ordering of the centres, fewer the one-launch rank sort: a spread-out
    sample (thousands of occupied bins) next to compact ones, in both positions"""
    rng = np.random.default_rng(12)
    wide = rng.uniform(-5, 5, size=(5000, 3)).astype(np.float32)
    cen = rng.normal(0, 3.0, size=(12, 3))
    tight = (cen[rng.integers(0, 12, size=2500)] + rng.normal(0, 0.1, size=(2500, 3))).astype(np.float32)
    for parts in ([tight, wide, tight[:900]], [tight, tight[:1200]]):
        x = np.concatenate(parts)
        offs = np.concatenate([[0], np.cumsum([len(q) for q in parts])]).tolist()
        wl, wn, _ = oracle.meanshift(x, offs, 0.6)
        labels, ncl, _ = ops.meanshift(dev(x), offs, 0.6)
        assert np.array_equal(ncl.cpu().numpy(), wn)
        assert np.array_equal(labels.cpu().numpy(), wl)


def test_hdbscan_matches_goldens(ops):
    z = np.load(os.path.join(GOLD, "hdbscan_cases.npz"))
    for name in z["names"].tolist():
        x = z["x_" + name]
        labels, ncl = ops.hdbscan(dev(x), [0, len(x)], 15, 5, float(z["eps_" + name]), count_self=True)
        assert np.array_equal(labels.cpu().numpy(), z["canon_" + name]), name   # sklearn's tree code, canonical tie order
        assert int(ncl[0]) == z["canon_" + name].max() + 1


@pytest.mark.parametrize("count_self", [True, False])
def test_hdbscan_batched_matches_oracle(ops, oracle, count_self):
    rng = np.random.default_rng(21)
    xs, offs = [], [0]
    for n, k, dim_sigma in [(1500, 9, 0.2), (3, 1, 0.1), (2600, 20, 0.1), (0, 1, 0.1), (40, 1, 0.3), (700, 3, 0.5), (5, 1, 0.1)]:
        cen = rng.normal(0, 3.0, size=(k, 5))
        x = cen[rng.integers(0, k, size=n)] + rng.normal(0, dim_sigma, size=(n, 5))
        if n > 100:
            x[: n // 20] = rng.uniform(-8, 8, size=(n // 20, 5))     # background noise
        xs.append(x.astype(np.float32))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    for eps in [0.006, 0.4]:
        wl, wn = oracle.hdbscan(x, offs, 15, 5, eps, count_self)
        labels, ncl = ops.hdbscan(dev(x), offs, 15, 5, eps, count_self=count_self)
        assert np.array_equal(ncl.cpu().numpy(), wn)
        assert np.array_equal(labels.cpu().numpy(), wl)
    assert wn[0] >= 5 and wn[2] >= 10 and wn[1] == 0 and wn[4] == 0  # one blob: no split -> noise (allow_single_cluster=False)


def test_hdbscan_lattice_ties_and_duplicates(ops, oracle):
    """Voxel-lattice coordinates (many exactly equal distances) and duplicated points: ties everywhere."""
    rng = np.random.default_rng(22)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    a = g[rng.random(len(g)) < 0.7] * 0.05
    b = a[: len(a) // 2] + np.array([3.0, 0, 0])
    dup = np.repeat(np.array([[9.0, 9, 9]]), 20, 0)
    x = np.concatenate([a, b, dup]).astype(np.float32)
    x = x[rng.permutation(len(x))]
    wl, wn = oracle.hdbscan(x, [0, len(x)], 15, 5, 0.006, True)
    labels, ncl = ops.hdbscan(dev(x), [0, len(x)], 15, 5, 0.006, count_self=True)
    assert np.array_equal(labels.cpu().numpy(), wl) and int(ncl[0]) == wn[0] and wn[0] >= 2


def test_hdbscan_wrapper_matches_reference_conventions(ops, oracle):
    from panopticsegforlargescalepointcloud_amd.utils import hdbscan_cluster as hc
    rng = np.random.default_rng(23)
    n = 1200
    batch = np.sort(rng.integers(0, 3, size=n))
    cen = rng.normal(0, 3, size=(8, 5))
    emb = (cen[rng.integers(0, 8, size=n)] + rng.normal(0, 0.1, size=(n, 5))).astype(np.float32)
    local = rng.permutation(5000)[:n]
    clusters, types = hc.cluster_single(dev(emb), None, dev(batch), dev(local), 1)
    offs = [0] + np.cumsum(np.bincount(batch, minlength=3)).tolist()
    wl, wn = oracle.hdbscan(emb, offs, 15, 5, 0.006, hc.COUNT_SELF)
    want = []
    for s in range(3):
        for l in range(wn[s]):
            want.append(local[offs[s]: offs[s + 1]][wl[offs[s]: offs[s + 1]] == l])
    assert len(clusters) == len(want) and types == [1] * len(want)
    assert all(np.array_equal(c.cpu().numpy(), w) for c, w in zip(clusters, want))
    cl2, t2 = hc.cluster_loop(dev(emb), None, dev(batch), dev(local), 3, 5, 2)
    assert len(cl2) == len(t2) and set(t2) <= {0, 1} and all(c.numel() >= 15 for c in cl2)


def test_group_by_key_and_segment_reduce(ops, oracle):
    rng = np.random.default_rng(11)
    key = rng.integers(-1, 50, size=10000).astype(np.int32)
    key[key == 7] = 8
    ids = rng.integers(0, 10 ** 6, size=10000)
    woffs, wout = oracle.group_by_key(key, 50, ids)
    offs, out, total = ops.group_by_key(dev(key), 50, dev(ids))
    assert np.array_equal(offs.cpu().numpy(), woffs)
    assert np.array_equal(out[: ops.group_by_key_check(total)].cpu().numpy(), wout)
    bad = key.copy()
    bad[3] = 50  # a key >= n_groups is a caller error and must be reported, not dropped silently
    with pytest.raises(Exception, match="outside"):
        ops.group_by_key_check(ops.group_by_key(dev(bad), 50, dev(ids))[2])
    src = rng.normal(size=(10000, 16)).astype(np.float32)
    index = rng.integers(0, 300, size=10000)
    index[index == 5] = 6
    for red in ["sum", "mean", "max"]:
        want, warg = oracle.segment_reduce(src, index, 300, red)
        got, arg = ops.segment_reduce(dev(src), dev(index), 300, red, want_arg=True)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
        if red == "max":
            assert np.array_equal(arg.cpu().numpy(), warg)


@pytest.mark.parametrize("n,c,n_seg", [(200_000, 5, 40), (30_000, 16, 3000), (5000, 1, 7), (70_000, 1, 12000),
                                       (100_003, 16, 5000), (9001, 4, 2000)])
def test_segment_reduce_few_and_many_segments(ops, oracle, n, c, n_seg):
    """both accumulation paths (block-private LDS for few segments, global atomics otherwise), empty segments and the
    arg-max convention; sums differ from the oracle only by float32 summation order."""
    rng = np.random.default_rng(n + c)
    src = rng.normal(size=(n, c)).astype(np.float32)
    index = rng.integers(0, n_seg, size=n)
    index[index == 3] = 4
    for order in ("random", "sorted"):  # sorted ids: the run-merging kernel takes one atomic per run
        if order == "sorted":
            index = np.sort(index)
        for red in ["sum", "mean", "max"]:
            want, warg = oracle.segment_reduce(src, index, n_seg, red)
            got, arg = ops.segment_reduce(dev(src), dev(index), n_seg, red, want_arg=True)
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-3 if red == "sum" else 2e-4)
            if red == "max":
                assert np.array_equal(arg.cpu().numpy(), warg)
    with pytest.raises(Exception):
        index[17] = n_seg
        ops.segment_reduce(dev(src), dev(index), n_seg, "sum")


def test_segment_reduce_run_structured_index(ops):
    """Morton-ordered points give long runs of equal segment ids (whole waves adding into one output address): the
    shapes of the discriminative loss, repeated; sums against float64."""
    rng = np.random.default_rng(99)
    for rep in range(12):
        n = int(rng.integers(15_000, 30_000))
        n_seg = int(rng.integers(3, 24))
        runs = rng.integers(1, 400, size=n)
        index = np.repeat(rng.integers(0, n_seg, size=n), runs)[:n]
        index[[1638, 3276, 8192]] = n_seg  # rows whose 5 columns straddle two thread blocks, alone in their segment
        n_seg += 1
        for c in (5, 1):
            src = (rng.normal(size=(n, c)) * 3).astype(np.float32)
            want = np.zeros((n_seg, c))
            np.add.at(want, index, src.astype(np.float64))
            got = ops.segment_reduce(dev(src), dev(index), n_seg, "sum")
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=5e-3)
            cnt = np.maximum(np.bincount(index, minlength=n_seg), 1).reshape(-1, 1)
            got = ops.segment_reduce(dev(src), dev(index), n_seg, "mean")
            np.testing.assert_allclose(got.cpu().numpy(), want / cnt, rtol=1e-5, atol=1e-4)


def test_gather_rows_matches_indexing(ops):
    rng = np.random.default_rng(21)
    for n_src, n, c in [(1000, 5000, 4), (70_000, 200_000, 16), (3000, 10, 112), (5, 0, 8), (4000, 4000, 64)]:
        src = dev(rng.normal(size=(n_src, c)).astype(np.float32))
        idx = dev(rng.integers(0, n_src, size=n))
        assert torch.equal(ops.gather_rows(src, idx), src[idx])
    # layouts the kernel does not take fall back to torch indexing (same result)
    src = dev(rng.normal(size=(100, 6)).astype(np.float32))
    idx = dev(rng.integers(0, 100, size=50))
    assert torch.equal(ops.gather_rows(src, idx), src[idx])
    ops.gather_rows_check()
    src = dev(rng.normal(size=(100, 8)).astype(np.float32))
    ops.gather_rows(src, dev(np.array([1, 100, -1, 5])))
    with pytest.raises(Exception):
        ops.gather_rows_check()
    ops.gather_rows_check()  # the flag is cleared by the failed check


def test_instance_iou_and_intersections_match_goldens(ops, oracle):
    z = np.load(os.path.join(GOLD, "loss_cases.npz"))
    offs = z["cluster_offsets"]
    clusters = [torch.from_numpy(z["cluster_points"][offs[i]: offs[i + 1]]) for i in range(len(offs) - 1)]
    csr = ops.ClusterCSR.from_list(clusters, "cuda")
    gt_off, gt_sizes = oracle.gt_layout(z["inst"], z["batch"])
    iou = ops.instance_iou_csr(csr, dev(z["inst"]), dev(z["batch"]), dev(gt_off), dev(gt_sizes))
    np.testing.assert_allclose(iou.cpu().numpy(), z["ious"], rtol=1e-6, atol=1e-7)
    z = np.load(os.path.join(GOLD, "nms_cases.npz"))
    offs = z["cluster_offsets"]
    clusters = [torch.from_numpy(z["cluster_points"][offs[i]: offs[i + 1]]) for i in range(len(offs) - 1)]
    csr = ops.ClusterCSR.from_list(clusters, "cuda")
    inter = ops.proposal_intersections(csr, int(z["n"]))
    want = oracle.proposal_intersections([c.numpy() for c in clusters], int(z["n"]))
    assert np.array_equal(inter.cpu().numpy(), want)


def test_morton_order_and_derived_maps(ops, oracle):
    rng = np.random.default_rng(12)
    fine = surface(rng, n=3000, n_batch=3, extent=60)
    d = dev(fine)
    perm = ops.morton_order(d).cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(len(fine)))
    # batch-major, Z-order inside a batch
    def key(c):
        k = 0
        for b in range(16):
            for a, v in enumerate((c[1] + 32768, c[2] + 32768, c[3] + 32768)):
                k |= ((int(v) >> b) & 1) << (3 * b + a)
        return (int(c[0]) << 48) | k
    keys = [key(c) for c in fine[perm]]
    assert keys == sorted(keys)
    # parity-grouped block order (unit = tensor stride 2, blocks of 2^3 units)
    coarse2 = np.unique(np.concatenate([fine[:, :1], fine[:, 1:] // 2 * 2], 1), axis=0).astype(np.int32)
    p2 = ops.morton_order(dev(coarse2), 2, 3).cpu().numpy()
    assert np.array_equal(np.sort(p2), np.arange(len(coarse2)))

    def key2(c, B=3):
        q = [(int(v) + 32768) >> 1 for v in c[1:]]
        z = lambda vals, nb: sum(((vals[a] >> b) & 1) << (3 * b + a) for b in range(nb) for a in range(3))  # noqa: E731
        inner = z([(v >> 1) & ((1 << (B - 1)) - 1) for v in q], B - 1)
        par = (q[0] & 1) | ((q[1] & 1) << 1) | ((q[2] & 1) << 2)
        return (int(c[0]) << 48) | (z([v >> B for v in q], 16 - B) << (3 * B)) | (par << (3 * (B - 1))) | inner
    keys2 = [key2(c) for c in coarse2[p2]]
    assert keys2 == sorted(keys2) and len(set(keys2)) == len(keys2)
    # coords[perm] decoded from the sorted keys == gathered, for every layout and unit
    for unit, bb, arr in [(1, 0, fine), (1, 4, fine), (2, 3, coarse2), (2, 5, coarse2)]:
        pm, srt = ops.morton_order(dev(arr), unit, bb, want_sorted=True)
        assert np.array_equal(srt.cpu().numpy(), arr[pm.cpu().numpy()])
    # derived maps == probed maps
    table, _ = ops.hash_build(d)
    coarse, ctable, _ = ops.stride_coords(d, 2)
    down = ops.kernel_map(coarse, table, 3, 1, 1)
    up_probe = ops.kernel_map(d, ctable, 3, 1, -1)
    up_derived = ops.kernel_map_transpose(down, len(fine))
    assert torch.equal(up_probe, up_derived)
    same = ops.kernel_map(d, table, 3, 1, 1)
    assert torch.equal(torch.flip(same, [0]), ops.kernel_map(d, table, 3, 1, -1))


@pytest.mark.parametrize("n", [0, 1, 63, 2047, 2048, 2049, 4096, 4097, 100_003, 5_000_000, 17_000_001])
def test_exclusive_scan(ops, n):
    """the library's own device scan (reduce-then-scan over 4096-element tiles; 17 M elements take two levels)"""
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda", generator=g)
    out, total = ops.exclusive_scan(x, want_total=True)
    want = torch.cumsum(x.long(), 0) - x.long()
    assert torch.equal(out.long(), want) and int(total) == int(x.long().sum())
    if n:  # in place
        y = x.clone()
        lib = ops._lib.load()
        wsb = lib.pp_exclusive_scan_workspace(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        ops._lib.check(lib.pp_exclusive_scan(y.data_ptr(), y.data_ptr(), n, None, ws.data_ptr(), wsb, None), "scan")
        assert torch.equal(y.long(), want)


@pytest.mark.parametrize("n,dtype,end_bit", [(0, torch.int32, 32), (1, torch.int64, 64), (2047, torch.int32, 32), (2049, torch.int64, 64),
                                              (100_003, torch.int32, 9), (100_003, torch.int64, 40), (3_000_000, torch.int64, 64),
                                              (3_000_000, torch.int32, 20), (70_001, torch.int32, 0)])
def test_sort_pairs(ops, n, dtype, end_bit):
    """the library's own stable LSD radix sort: sorted by the low end_bit bits only, ties keep their input order"""
    g = torch.Generator(device="cuda").manual_seed(n + end_bit)
    hi = 2 ** 31 - 1 if dtype == torch.int32 else 2 ** 62
    keys = torch.randint(0, hi, (n,), dtype=dtype, device="cuda", generator=g)
    if n > 10:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]  # plenty of ties
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo = ops.sort_pairs(keys, vals, end_bit)
    masked = keys.long() & ((1 << end_bit) - 1) if end_bit < 63 else keys.long()
    order = torch.sort(masked, stable=True)[1]
    assert torch.equal(vo.long(), order) and torch.equal(ko, keys[order])


@pytest.mark.parametrize("n", [1, 2047, 2049, 300_001])
def test_sort_pairs_with_constant_digits(ops, n):
    """keys whose bytes 1, 3 and 5..7 are the same in every key (Morton keys of a small scene look like this): passes that are
    the identity permutation between passes that are not; the result is the same stable sort"""
    g = torch.Generator(device="cuda").manual_seed(n)
    lo = torch.randint(0, 256, (n,), dtype=torch.int64, device="cuda", generator=g)
    mid = torch.randint(0, 256, (n,), dtype=torch.int64, device="cuda", generator=g)
    hi = torch.randint(0, 3, (n,), dtype=torch.int64, device="cuda", generator=g)
    keys = lo | (0x5A << 8) | (mid << 16) | (0x11 << 24) | (hi << 32) | (0x0102_03 << 40)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo = ops.sort_pairs(keys, vals, 64)
    order = torch.sort(keys, stable=True)[1]
    assert torch.equal(vo.long(), order) and torch.equal(ko, keys[order])
    k32 = (lo | (0x33 << 8) | (mid << 16)).to(torch.int32)
    ko, vo = ops.sort_pairs(k32, vals, 32)
    order = torch.sort(k32.long(), stable=True)[1]
    assert torch.equal(vo.long(), order) and torch.equal(ko, k32[order])


def _mask_sort_rank(mask, window):
    """numpy restatement of the sort key of csrc/pp_maporder.hip: per window of `window` consecutive rows the offset most rows
    have is the least significant bit, the rarest the most significant (ties: lower offset index lower).  `mask` in the
    ORIGINAL row order; returns the keys in that order."""
    out = np.zeros(len(mask), np.int64)
    for w0 in range(0, len(mask), window):
        m = mask[w0: w0 + window]
        freq = np.array([int(((m >> k) & 1).sum()) for k in range(27)])
        pos = [sum(1 for j in range(27) if freq[j] > freq[k] or (freq[j] == freq[k] and j < k)) for k in range(27)]
        o = np.zeros(len(m), np.int64)
        for k in range(27):
            o |= ((m >> k) & 1) << pos[k]
        out[w0: w0 + window] = o
    return out


def test_map_order_and_slot_ordered_maps(ops, oracle):
    """pp_map_order is a window-local permutation sorted by (remapped neighbour mask, row); pp_map_permute /
    pp_level_permute / pp_kernel_map_transpose(order) restate the map in slot order; pp_spconv_fwd on the slot-ordered
    map + row_order is bit-identical to the plain map (same per-row summation order), incl. cat / BN / ReLU / residual."""
    rng = np.random.default_rng(32)
    lib_window = ops._lib.load().pp_map_window()
    for n_pts, n_batch in [(24000, 3), (300, 2), (1, 1)]:
        fine = surface(rng, n=n_pts, n_batch=n_batch, extent=110)
        fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
        n = len(fine)
        idx, _ = ops.block_index_build(dev(fine), 1, 4)
        nbr = ops.kernel_map_bi(dev(fine), idx, 3, 1, 1, want_mask=True)
        mask = nbr.pp_mask.cpu().numpy().astype(np.int64)
        want_mask = ((nbr.cpu().numpy() >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
        assert np.array_equal(mask, want_mask)
        assert np.array_equal(ops.map_mask(nbr).cpu().numpy().astype(np.int64), want_mask)
        order = ops.map_order(nbr.pp_mask)
        o = order.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(o), np.arange(n))
        assert np.array_equal(o // lib_window, np.arange(n) // lib_window)          # rows never leave their window
        key = ((np.arange(n) // lib_window) << 50) | (_mask_sort_rank(mask, lib_window)[o] << 13) | (o % lib_window)
        assert np.all(np.diff(key) > 0)
        # renumbered level + same-level map in physical ids
        coords_p, phys_of = ops.level_permute(dev(fine), order)
        assert np.array_equal(coords_p.cpu().numpy(), fine[o])
        inv = np.empty(n, np.int64)
        inv[o] = np.arange(n)
        assert np.array_equal(phys_of.cpu().numpy(), inv)
        same = ops.map_permute(nbr, order, translate=phys_of).cpu().numpy()
        assert np.array_equal(same, oracle.kernel_map(fine[o], fine[o], 3, 1, 1))
        if n < 100:
            continue
        # cross-level maps: slot-major + order; transposed map from the slot-ordered strided one
        coarse = np.unique(np.concatenate([fine[:, :1], fine[:, 1:] // 2 * 2], 1), axis=0).astype(np.int32)
        coarse = coarse[ops.morton_order(dev(coarse), 2, 4).cpu().numpy()]
        down = ops.kernel_map_bi(dev(coarse), idx, 3, 1, 1, want_mask=True)      # coarse rows gather fine (block-order) rows
        od = ops.map_order(down.pp_mask)
        down_s = ops.map_permute(down, od, translate=phys_of)
        down_t = ops.kernel_map_bi(dev(coarse), idx, 3, 1, 1, translate=phys_of)  # translation folded into the lookup
        assert torch.equal(down_t, ops.map_permute(down, None, translate=phys_of))
        od_np = od.cpu().numpy()
        assert np.array_equal(down_s.cpu().numpy(), oracle.kernel_map(coarse[od_np], fine[o], 3, 1, 1))
        up = ops.kernel_map_transpose(down_s, n, order=od)                        # rows = fine physical rows, values = coarse rows
        assert np.array_equal(up.cpu().numpy(), oracle.kernel_map(fine[o], coarse, 3, 1, -1))
        ou = ops.map_order(ops.map_mask(up))
        up_s = ops.map_permute(up, ou)
        for (m_plain, m_slot, m_order, n_in, n_out, cin, cout, c1) in [
                (dev(same), dev(same), None, n, n, 16, 16, 0),
                (ops.map_permute(down, None, translate=phys_of), down_s, od, n, len(coarse), 32, 48, 32),
                (up, up_s, ou, len(coarse), n, 64, 64, 0), (up, up_s, ou, len(coarse), n, 96, 80, 0)]:
            x = rng.normal(size=(n_in, cin)).astype(np.float32)
            x1 = rng.normal(size=(n_in, c1)).astype(np.float32) if c1 else None
            w = (rng.normal(size=(27, cin + c1, cout)) * 0.1).astype(np.float32)
            res = rng.normal(size=(n_out, cout)).astype(np.float32)
            sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
            kw = dict(in1=None if x1 is None else dev(x1), scale=dev(sc), shift=dev(sh), relu=True, residual=dev(res))
            a = ops.spconv_fwd(dev(x), ops.pack_weight(dev(w)), m_plain, n_out, cout, 27, **kw)
            b = ops.spconv_fwd(dev(x), ops.pack_weight(dev(w)), m_slot, n_out, cout, 27, row_order=m_order, **kw)
            assert torch.equal(a, b)
            want = oracle.spconv_fwd(x, w, m_plain.cpu().numpy(), n_out, in1=x1, scale=sc, shift=sh, relu=True, residual=res)
            np.testing.assert_allclose(b.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("window", [1024, 2048, 8192, 16384, 32768])
def test_map_order_large_windows(ops, oracle, window):
    """pp_map_order_window with every window size it takes (8192 / 16384 are the levels' own orders, the others A/B settings;
    4096 is refused, test_map_order_beside_another_stream): a window-local permutation sorted by
    (remapped neighbour mask, row); pp_map_permute with that window restates the same-level map in the new row ids."""
    rng = np.random.default_rng(33)
    for n_pts, n_batch in [(70000, 2), (window // 30, 1)]:
        fine = surface(rng, n=n_pts, n_batch=n_batch, extent=160)
        fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
        n = len(fine)
        idx, _ = ops.block_index_build(dev(fine), 1, 4)
        nbr = ops.kernel_map_bi(dev(fine), idx, 3, 1, 1, want_mask=True)
        mask = nbr.pp_mask.cpu().numpy().astype(np.int64)
        order = ops.map_order(nbr.pp_mask, window=window)
        assert order.pp_window == window
        o = order.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(o), np.arange(n))
        assert np.array_equal(o // window, np.arange(n) // window)
        key = ((np.arange(n) // window) << 50) | (_mask_sort_rank(mask, window)[o] << 15) | (o % window)
        assert np.all(np.diff(key) > 0)
        coords_p, phys_of = ops.level_permute(dev(fine), order)
        same = ops.map_permute(nbr, order, translate=phys_of).cpu().numpy()
        assert np.array_equal(same, oracle.kernel_map(fine[o], fine[o], 3, 1, 1))
        plain = ops.map_permute(nbr, order).cpu().numpy()                       # no translation: old row ids
        assert np.array_equal(plain, nbr.cpu().numpy()[:, o])


@pytest.mark.parametrize("window", [2048, 8192, 16384])
def test_map_order_beside_another_stream(ops, window):
    """pp_map_order_window / pp_map_permute on one stream while a second thread keeps another stream busy with the same kind of
    work (what the coordinate manager's early prefetch does beside the constructor): every order stays a window-local permutation
    and every permuted map the plain gather of the map.  The 4096-row window did NOT pass this (rows left their window beside
    k_kernel_map_bi / k_map_permute_big on the other stream -- the memory faults of the bench's window sweeps in rounds 4 and 6,
    profiles/r06_sort_race_probe.txt) and is refused by the library since."""
    with pytest.raises(Exception):
        ops.map_order(torch.zeros(10000, dtype=torch.int32, device="cuda"), window=4096)
    import threading
    rng = np.random.default_rng(61)
    fine = surface(rng, n=400000, n_batch=4, extent=900)
    fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
    n = len(fine)
    idx, _ = ops.block_index_build(dev(fine), 1, 4)
    nbr = ops.kernel_map_bi(dev(fine), idx, 3, 1, 1, want_mask=True)
    mask = nbr.pp_mask
    torch.cuda.synchronize()
    stop = threading.Event()
    side = torch.cuda.Stream()
    err = []

    def noise():
        try:
            with torch.cuda.stream(side):
                while not stop.is_set():
                    cidx, cc = ops.block_index_coarsen(idx, n)
                    m2 = ops.kernel_map_bi(cc, cidx, 3, 2, 1, want_mask=True)
                    o2 = ops.map_order(m2.pp_mask, window=16384)
                    c2, p2 = ops.level_permute(cc, o2)
                    ops.map_permute(m2, o2, translate=p2)
        except BaseException as e:  # noqa: BLE001
            err.append(e)

    th = threading.Thread(target=noise)
    th.start()
    try:
        ar = torch.arange(n, device="cuda")
        for it in range(25):
            order = ops.map_order(mask, window=window)
            o = order.long()
            assert torch.equal(o // window, ar // window), "iteration %d: rows left their window" % it
            assert torch.equal(torch.sort(o)[0], ar), "iteration %d: not a permutation" % it
            coords_p, phys_of = ops.level_permute(dev(fine), order)
            plain = ops.map_permute(nbr, order)
            assert torch.equal(plain, nbr[:, o]), it
    finally:
        stop.set()
        th.join()
    assert not err, err


def test_proposals_unique_front_end(ops):
    """pp_proposals_unique / pp_proposals_emit against a dictionary of point lists: representatives are the smallest index of
    an IDENTICAL list (same points, same order -- a permuted copy has the same hashes and must be kept), kept lists / batch
    index / coordinate rows are those of the representatives in index order; empty input and a single proposal included."""
    rng = np.random.default_rng(41)
    n_points = 5000
    coords = rng.integers(-300, 300, size=(n_points, 3)).astype(np.int32)
    for n_prop in [0, 1, 7, 600]:
        lists, fresh, flipped = [], [], set()
        for p in range(n_prop):
            r = rng.random()
            if fresh and r < 0.3:
                lists.append(lists[fresh[rng.integers(len(fresh))]].copy())           # exact duplicate of an ascending list
            elif fresh and r < 0.4 and len(lists[fresh[-1]]) > 1 and fresh[-1] not in flipped:
                # same set, other order: same hashes, NOT a duplicate.  (A copy of such a list would be compared with the
                # ascending one -- the smallest index under the signature -- and kept as well: exact, just not deduplicated.)
                lists.append(lists[fresh[-1]][::-1].copy())
                flipped.add(fresh[-1])
            else:
                lists.append(np.sort(rng.choice(n_points, size=int(rng.integers(1, 400)), replace=False)).astype(np.int64))
                fresh.append(p)
        csr = ops.ClusterCSR.from_list([torch.from_numpy(l) for l in lists], torch.device("cuda")) if n_prop else \
            ops.ClusterCSR(torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 0)
        u = ops.proposals_unique(csr, n_points, coords=dev(coords))
        first, rep = {}, []
        for p, l in enumerate(lists):
            rep.append(first.setdefault(l.tobytes(), p))
        kept = [p for p in range(n_prop) if rep[p] == p]
        assert u.csr.n == len(kept)
        want_offs = np.concatenate([[0], np.cumsum([len(lists[p]) for p in kept])]).astype(np.int32)
        assert np.array_equal(u.csr.offsets.cpu().numpy(), want_offs)
        want_pts = np.concatenate([lists[p] for p in kept]) if kept else np.zeros(0, np.int64)
        assert np.array_equal(u.csr.points.cpu().numpy(), want_pts)
        pos = {p: i for i, p in enumerate(kept)}
        assert np.array_equal(u.pos_of.cpu().numpy(), np.array([pos[r] for r in rep], np.int64))
        want_b = np.repeat(np.arange(len(kept)), [len(lists[p]) for p in kept])
        assert np.array_equal(u.batch.cpu().numpy(), want_b)
        assert np.array_equal(u.coords4.cpu().numpy(), np.concatenate([want_b[:, None].astype(np.int32), coords[want_pts]], 1))
    bad = ops.ClusterCSR.from_list([torch.tensor([1, n_points])], torch.device("cuda"))
    with pytest.raises(ops._lib.PanopticHipError):
        ops.proposals_unique(bad, n_points)


def test_segment_sum_without_atomics_matches_and_repeats(ops):
    """pp_segment_sum_ordered (the default of ops.segment_reduce for sum / mean): equal to a float64 reference within float
    rounding, bit-identical over repeated calls (the atomic kernels are not), empty segments 0, out-of-range ids reported."""
    rng = np.random.default_rng(43)
    for n, c, n_seg in [(50000, 5, 37), (3000, 16, 2000), (70000, 300, 3), (0, 4, 5)]:
        x = rng.normal(size=(n, c)).astype(np.float32) * 10
        idx = rng.integers(0, n_seg, size=n)
        if n_seg > 3:
            idx[idx == 2] = 3                                            # an empty segment
        for reduce in ("sum", "mean"):
            want = np.zeros((n_seg, c))
            np.add.at(want, idx, x.astype(np.float64))
            if reduce == "mean":
                want /= np.maximum(np.bincount(idx, minlength=n_seg), 1)[:, None]
            outs = [ops.segment_reduce(dev(x), dev(idx.astype(np.int64)), n_seg, reduce) for _ in range(3)]
            np.testing.assert_allclose(outs[0].cpu().numpy(), want, rtol=2e-5, atol=2e-3)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with pytest.raises(ops._lib.PanopticHipError):
        ops.segment_reduce(dev(np.ones((4, 2), np.float32)), dev(np.array([0, 1, 5, 1], np.int64)), 3, "sum")


def test_compact_same_level_map(ops, oracle):
    """pp_map_compact_* against numpy (mask, chunk offsets, entries offset-major inside 32-row chunks, tags) and
    pp_spconv_fwd_cmap bit-identical to pp_spconv_fwd on the dense map: 16 / 32 / 64 / 96+32 input channels (64 and 32 rows per
    wave, split-K on the small map, the 4-channel input layer's grouped loop, BN / ReLU / residual / fused shortcut), row counts
    that are not multiples of 32."""
    rng = np.random.default_rng(47)
    for n_pts, n_batch in [(60000, 2), (900, 1), (33, 1)]:
        fine = surface(rng, n=n_pts, n_batch=n_batch, extent=150)
        fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
        n = len(fine)
        idx, _ = ops.block_index_build(dev(fine), 1, 4)
        nbr = ops.kernel_map_bi(dev(fine), idx, 3, 1, 1)
        cm = ops.map_compact(nbr)
        d = nbr.cpu().numpy()
        present = d >= 0
        assert np.array_equal(cm.mask.cpu().numpy().astype(np.int64) & 0x7FFFFFF, (present.astype(np.int64) << np.arange(27)[:, None]).sum(0))
        chunks = (n + 31) // 32
        cnt = np.add.reduceat(present.sum(0), np.arange(0, n, 32))
        start = np.concatenate([[0], np.cumsum(cnt)])
        assert np.array_equal(cm.start.cpu().numpy(), start)
        P = int(start[-1])
        ent, tag = cm.entries[:P].cpu().numpy(), cm.tags[:P].cpu().numpy().astype(np.int64) & 0xFFFF
        want_ent, want_tag = [], []
        for c in range(chunks):
            blk = d[:, 32 * c: 32 * c + 32]
            k, r = np.nonzero(blk >= 0)                       # offset-major, rows ascending
            want_ent.append(blk[k, r])
            want_tag.append(k * 64 + ((32 * c + r) & 63))
        assert np.array_equal(ent, np.concatenate(want_ent)) and np.array_equal(tag, np.concatenate(want_tag))
        for c0, c1, cout, kw in [(16, 0, 16, {}), (32, 0, 32, {"relu": True}), (64, 0, 64, {}), (96, 32, 48, {}), (32, 32, 48, {}), (4, 0, 16, {}),
                                 (32, 0, 64, {"shortcut": 32})]:  # (96 + 32: not the pipelined kernel's shape -- dense map)
            x = dev(rng.normal(size=(n, c0)).astype(np.float32))
            x1 = dev(rng.normal(size=(n, c1)).astype(np.float32)) if c1 else None
            w = ops.pack_weight(dev((rng.normal(size=(27, c0 + c1, cout)) * 0.1).astype(np.float32)))
            sc, sh = dev(rng.uniform(0.5, 1.5, cout).astype(np.float32)), dev(rng.normal(size=cout).astype(np.float32))
            res = dev(rng.normal(size=(n, cout)).astype(np.float32))
            args = dict(in1=x1, scale=sc, shift=sh, relu=bool(kw.get("relu")), residual=res)
            if "shortcut" in kw:
                xs = dev(rng.normal(size=(n, kw["shortcut"])).astype(np.float32))
                ws = ops.pack_weight(dev((rng.normal(size=(1, kw["shortcut"], cout)) * 0.1).astype(np.float32)))
                args["shortcut"] = (xs, ws, sc, sh)
            plain = torch.empty_like(nbr).copy_(nbr)
            a = ops.spconv_fwd(x, w, plain, n, cout, 27, **args)
            nbr.pp_cmap = cm
            b = ops.spconv_fwd(x, w, nbr, n, cout, 27, **args)
            del nbr.pp_cmap
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b), (n, c0, c1, cout)


def test_linear_rows_matches_float64(ops):
    """pp_linear_rows (forward and input gradient of the heads' skinny Linear layers in training): y = x W^T + b and y = x W
    against float64, bias optional, channel counts that are not multiples of four, a row count that is not a multiple of 256."""
    rng = np.random.default_rng(53)
    for n, cin, cout in [(70001, 16, 16), (5000, 16, 3), (300, 5, 32), (1, 32, 1), (0, 16, 16)]:
        x = rng.normal(size=(n, cin)).astype(np.float32)
        w = rng.normal(size=(cout, cin)).astype(np.float32)
        b = rng.normal(size=cout).astype(np.float32)
        y = ops.linear_rows(dev(x), dev(w), dev(b)).cpu().numpy()
        np.testing.assert_allclose(y, x.astype(np.float64) @ w.T.astype(np.float64) + b, rtol=1e-5, atol=1e-5)
        y0 = ops.linear_rows(dev(x), dev(w)).cpu().numpy()
        np.testing.assert_allclose(y0, x.astype(np.float64) @ w.T.astype(np.float64), rtol=1e-5, atol=1e-5)
        dy = rng.normal(size=(n, cout)).astype(np.float32)
        dx = ops.linear_rows(dev(dy), dev(w), None, transposed=True).cpu().numpy()
        np.testing.assert_allclose(dx, dy.astype(np.float64) @ w.astype(np.float64), rtol=1e-5, atol=1e-5)


VARIANT_SHAPES = [("same", 16, 0, 16), ("same", 32, 32, 48), ("strided", 32, 0, 32), ("transposed", 64, 0, 64),
                  ("transposed", 48, 48, 32)]


@pytest.mark.parametrize("kind,c0,c1,cout", VARIANT_SHAPES)
@pytest.mark.parametrize("rows_per_wave", [32, 64])
@pytest.mark.parametrize("pipeline", [1, 3, 5, 6])
def test_spconv_kernel_variants_match_oracle(ops, oracle, kind, c0, c1, cout, rows_per_wave, pipeline):
    """Every variant of the pipelined kernel the benchmark selects by shape -- 32 / 64 rows per wave, the loop forms (1, 3:
    one step of operand loads in flight; 5: the depth-3 register ring with hand-counted waits; 6: LDS-staged feature tiles --
    full-line gathers by buffer_load ... lds into an XOR-swizzled ring, fragments by ds_read_b128), unsplit and split-K -- on >= 20 k-row same-level, strided and transposed maps with the fused second source (ME.cat),
    folded BN, ReLU and residual, in fp32 (1e-4 vs the oracle) and with bfloat16 compute (oracle on rounded operands)."""
    if pipeline == 6 and (c0 % 32 or c1 % 32 or rows_per_wave != 32):
        pytest.skip("the LDS-staged loop serves rows of >= 128 bytes (channel counts that are multiples of 32), 32 rows per wave")
    rng = np.random.default_rng(101)
    fine = surface(rng, n=26000, n_batch=3, extent=120)
    coarse, _ = oracle.stride_coords(fine, 2)
    out_c, in_c, sign = {"same": (fine, fine, 1), "strided": (coarse, fine, 1), "transposed": (fine, coarse, -1)}[kind]
    nbr = oracle.kernel_map(out_c, in_c, 3, 1, sign)
    n_in, n_out = len(in_c), len(out_c)
    assert n_out >= 10000 and max(n_in, n_out) >= 20000
    x0 = rng.normal(size=(n_in, c0)).astype(np.float32)
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32) if c1 else None
    W = (rng.normal(size=(27, c0 + c1, cout)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(n_out, cout)).astype(np.float32)
    packed = ops.pack_weight(dev(W))
    kw = dict(in1=None if x1 is None else dev(x1), scale=dev(sc), shift=dev(sh), relu=True, residual=dev(res))
    want = oracle.spconv_fwd(x0, W, nbr, n_out, in1=x1, scale=sc, shift=sh, relu=True, residual=res)
    rb = oracle.round_bf16
    want_bf = oracle.spconv_fwd(rb(x0), rb(W), nbr, n_out, in1=None if x1 is None else rb(x1), scale=sc, shift=sh, relu=True,
                                residual=res)
    outs = []
    for split in (1, 2):  # (split 2: the partial sums of 78 k rows x 64 channels fit the default 64 MiB scratch)
        got = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(rows_per_wave, pipeline, split), **kw)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
        outs.append(got)
        got_bf = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, bf16=True, variant=(rows_per_wave, pipeline, split),
                                **kw).cpu().numpy()
        np.testing.assert_allclose(got_bf, want_bf, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want_bf).max())))
    assert not torch.equal(outs[0], outs[1])  # split-K really took another summation order
    # the unsplit variants share one summation order per row: bit-identical across rows-per-wave / loop forms
    ref = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(32, 1, 1), **kw)
    assert torch.equal(ref, outs[0])


X3_SHAPES = [("same", 64, 0, 64), ("same", 48, 0, 48), ("same", 80, 80, 64), ("strided", 48, 0, 48), ("transposed", 96, 96, 80),
             ("transposed", 64, 64, 128), ("same", 32, 0, 64), ("same", 112, 0, 112),
             # two column tiles per wave: the 32-output-channel layers (k_spconv_x3<2, ...>, the template with the largest share of the
             # bench step) incl. a strided launch.  (The test surface has 4.4 pairs per row -- the density of the proposal scorer's levels, 4.6;
             # "dust" thins it to 2.4: most tiles then walk offsets only one or two of their rows have)
             ("same", 32, 0, 32), ("same", 64, 0, 32), ("same", 96, 0, 32), ("same", 32, 32, 32), ("strided", 32, 0, 32),
             ("dust", 32, 0, 32), ("dust", 64, 0, 32)]


@pytest.mark.parametrize("kind,c0,c1,cout", X3_SHAPES)
def test_spconv_x3_default_wide_layers(ops, oracle, kind, c0, c1, cout):
    """Layers with >= 2 column tiles per wave (>= 32 output channels; on two tiles >= 32 input channels) run by default on
    k_spconv_x3 (pp_spconv3.hip): fp32 operands split EXACTLY into three
    bfloat16 terms, six bf16 MFMA products, fp32 accumulation.  Against the fp32 oracle at 1e-4 like every other variant; against
    a float64 evaluation of the same sums its error must not exceed the fp32-MFMA kernel's (forced through variant=(32, 1, 1))
    by more than a rounding or two -- it is an fp32-accurate evaluation, not a reduced-precision one; odd numbers of 16-channel
    steps (48, 80, 112: the last group is half empty), two sources whose boundary falls inside a group (80 + 80), strided,
    transposed and split-K launches, all with folded BN, ReLU and residual."""
    rng = np.random.default_rng(7)
    fine = surface(rng, n=26000, n_batch=3, extent=120)
    if kind == "dust":  # a thinned surface: most neighbours are missing, a 16-row tile walks many offsets for few pairs
        fine = fine[np.sort(rng.choice(len(fine), int(0.42 * len(fine)), replace=False))]
    coarse, _ = oracle.stride_coords(fine, 2)
    out_c, in_c, sign = {"same": (fine, fine, 1), "dust": (fine, fine, 1), "strided": (coarse, fine, 1),
                         "transposed": (fine, coarse, -1)}[kind]
    nbr = oracle.kernel_map(out_c, in_c, 3, 1, sign)
    n_in, n_out = len(in_c), len(out_c)
    if kind == "dust":
        ppr = float((nbr >= 0).sum()) / n_out
        assert 1.5 < ppr < 4.0, ppr
    x0 = (rng.normal(size=(n_in, c0)) * np.exp(rng.normal(size=(n_in, 1)))).astype(np.float32)  # rows of very different magnitude
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32) if c1 else None
    W = (rng.normal(size=(27, c0 + c1, cout)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(n_out, cout)).astype(np.float32)
    packed = ops.pack_weight(dev(W))
    kw = dict(in1=None if x1 is None else dev(x1), scale=dev(sc), shift=dev(sh), relu=True, residual=dev(res))
    want = oracle.spconv_fwd(x0, W, nbr, n_out, in1=x1, scale=sc, shift=sh, relu=True, residual=res)
    got = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, **kw)            # default: X3 on these shapes
    f32 = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(32, 1, 1), **kw)  # the fp32-MFMA kernel
    assert not torch.equal(got, f32), "the default path did not take the split-bf16 kernel"
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # float64 evaluation of the convolution sums (then the same fp32 epilogue arithmetic in float64)
    xin = x0 if x1 is None else np.concatenate([x0, x1], 1)
    ref = np.zeros((n_out, cout))
    for k in range(27):
        ok = nbr[k] >= 0
        ref[ok] += xin[nbr[k][ok]].astype(np.float64) @ W[k].astype(np.float64)
    ref = np.maximum(ref * sc.astype(np.float64) + sh.astype(np.float64), 0.0) + res.astype(np.float64)
    mag = np.abs(ref).max()
    e_x3 = float(np.abs(got.cpu().numpy() - ref).max() / mag)
    e_f32 = float(np.abs(f32.cpu().numpy() - ref).max() / mag)
    print("max error / magnitude vs float64: split-bf16 %.2e, fp32 MFMA %.2e" % (e_x3, e_f32))
    assert e_x3 <= max(2.0 * e_f32, 5e-7), (e_x3, e_f32)
    # split-K on the same kernel (another summation order, same accuracy), and run-to-run bit identity
    if 2 * n_out * cout * 4 <= ops.CONV_SCRATCH_BYTES:  # (the partial sums must fit the split-K scratch)
        sp = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(0, 0, 2), **kw)
        np.testing.assert_allclose(sp.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    again = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, **kw)
    assert torch.equal(got, again)
    # bfloat16 compute on the same kernel (one v_mfma_f32_16x16x32_bf16 per tile and 32 channels; weights rounded at packing
    # time, rows in registers): the fp32 oracle on operands rounded to nearest-even bfloat16 (their products are exact in fp32)
    rb = oracle.round_bf16
    want_bf = oracle.spconv_fwd(rb(x0), rb(W), nbr, n_out, in1=None if x1 is None else rb(x1), scale=sc, shift=sh, relu=True,
                                residual=res)
    got_bf = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, bf16=True, **kw)
    old_bf = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, bf16=True, variant=(32, 1, 1), **kw)  # k_spconv_fwd3
    assert not torch.equal(got_bf, old_bf)
    np.testing.assert_allclose(got_bf.cpu().numpy(), want_bf, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want_bf).max())))


@pytest.mark.parametrize("kind,c0,c1,cout", [("same", 64, 0, 64), ("same", 32, 0, 32), ("same", 96, 0, 48), ("same", 64, 64, 32),
                                             ("strided", 32, 0, 96), ("transposed", 64, 64, 80), ("dust", 128, 0, 64),
                                             ("same", 160, 0, 96)])
def test_spconv_x3_full_line_gathers_are_bit_identical(ops, oracle, kind, c0, c1, cout):
    """k_spconv_x3f (rows gathered as whole 128-byte lines straight into LDS, fragments read back from LDS, no neighbour table)
    against k_spconv_x3 (register gathers in fragment shape): same packed weights, same summation order -> the same bits, on dense
    maps of every kind, two sources, odd row counts (the 16-byte map loads are then only 4-byte aligned; the last workgroup has
    empty slots and whole empty waves), the fused shortcut, a permuted output order, split-K and bfloat16 compute; and both against
    the oracle."""
    rng = np.random.default_rng(23)
    fine = surface(rng, n=26000, n_batch=3, extent=120)
    if kind == "dust":
        fine = fine[np.sort(rng.choice(len(fine), int(0.42 * len(fine)), replace=False))]
    fine = fine[: len(fine) - (len(fine) % 128) - 37]  # n_out % 128 = 91: a ragged last workgroup, one wave without rows
    coarse, _ = oracle.stride_coords(fine, 2)
    out_c, in_c, sign = {"same": (fine, fine, 1), "dust": (fine, fine, 1), "strided": (coarse, fine, 1),
                         "transposed": (fine, coarse, -1)}[kind]
    nbr = oracle.kernel_map(out_c, in_c, 3, 1, sign)
    n_in, n_out = len(in_c), len(out_c)
    x0 = (rng.normal(size=(n_in, c0)) * np.exp(rng.normal(size=(n_in, 1)))).astype(np.float32)
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32) if c1 else None
    W = (rng.normal(size=(27, c0 + c1, cout)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(n_out, cout)).astype(np.float32)
    packed = ops.pack_weight(dev(W))
    kw = dict(in1=None if x1 is None else dev(x1), scale=dev(sc), shift=dev(sh), relu=True, residual=dev(res))
    want = oracle.spconv_fwd(x0, W, nbr, n_out, in1=x1, scale=sc, shift=sh, relu=True, residual=res)
    perm = rng.permutation(n_out).astype(np.int32)
    nbr_p = np.ascontiguousarray(nbr[:, perm])

    def run_all():
        out = [ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, **kw),
               ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, bf16=True, **kw),
               ops.spconv_fwd(dev(x0), packed, dev(nbr_p), n_out, cout, 27, row_order=dev(perm), **kw)]
        if 2 * n_out * cout * 4 <= ops.CONV_SCRATCH_BYTES:
            out.append(ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(0, 0, 2), **kw))
        if kind in ("same", "dust") and c1 == 0 and cout <= 64:
            xs = rng2.normal(size=(n_out, 32)).astype(np.float32)
            W1 = (rng2.normal(size=(1, 32, cout)) * 0.1).astype(np.float32)
            out.append(ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, scale=dev(sc), shift=dev(sh), relu=True,
                                      shortcut=(dev(xs), ops.pack_weight(dev(W1)), dev(sc), dev(sh))))
        return out

    was = ops.spconv_x3_full_lines(1)
    try:
        rng2 = np.random.default_rng(5)
        lines = run_all()
        ops.spconv_x3_full_lines(0)
        rng2 = np.random.default_rng(5)
        frags = run_all()
    finally:
        ops.spconv_x3_full_lines(was)
    f32 = ops.spconv_fwd(dev(x0), packed, dev(nbr), n_out, cout, 27, variant=(32, 1, 1), **kw)
    assert not torch.equal(lines[0], f32), "the default path did not take the split-bf16 kernel"
    np.testing.assert_allclose(lines[0].cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert torch.equal(lines[2], lines[0])  # a row's bits do not depend on its workgroup
    for a_, b_ in zip(lines, frags):  # (a launch small enough for split-K declines the fused shortcut: None both times)
        assert (a_ is None) == (b_ is None) and (a_ is None or torch.equal(a_, b_))


def test_spconv_x3_fused_shortcut_and_row_subsets(ops, oracle):
    """the fused 1x1 shortcut on the split-bf16 kernel == its two launches bit for bit, and a row's result does not depend on
    which other rows share its workgroup (the workgroup walks the union of its rows' offsets in lockstep; a row's own sum
    only contains its own offsets, in ascending order)"""
    rng = np.random.default_rng(11)
    fine = surface(rng, n=70000, n_batch=2, extent=200)  # (large enough that the launch is not split over the offsets)
    nbr = oracle.kernel_map(fine, fine, 3, 1, 1)
    n = len(fine)
    cin, cout, cs = 64, 64, 32
    x = rng.normal(size=(n, cin)).astype(np.float32)
    xs = rng.normal(size=(n, cs)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) * 0.1).astype(np.float32)
    W1 = (rng.normal(size=(1, cs, cout)) * 0.1).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    sc1, sh1 = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    pk, pk1 = ops.pack_weight(dev(W)), ops.pack_weight(dev(W1))
    short = ops.spconv_fwd(dev(xs), pk1, None, n, cout, 1, scale=dev(sc1), shift=dev(sh1))
    two = ops.spconv_fwd(dev(x), pk, dev(nbr), n, cout, 27, scale=dev(sc), shift=dev(sh), relu=True, residual=short)
    one = ops.spconv_fwd(dev(x), pk, dev(nbr), n, cout, 27, scale=dev(sc), shift=dev(sh), relu=True,
                         shortcut=(dev(xs), pk1, dev(sc1), dev(sh1)))
    assert one is not None
    want = oracle.spconv_fwd(x, W, nbr, n, scale=sc, shift=sh, relu=True) + (xs @ W1[0]) * sc1 + sh1
    np.testing.assert_allclose(one.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert torch.equal(two, one)  # (1x1 convolutions stay on the fp32 MFMAs in both forms)
    # a permuted OUTPUT order (row_order) puts every row into another workgroup: same bits per row
    perm = rng.permutation(n).astype(np.int32)
    nbr_p = np.ascontiguousarray(nbr[:, perm])  # slot-major map: slot s computes output row perm[s]
    shuf = ops.spconv_fwd(dev(x), pk, dev(nbr_p), n, cout, 27, scale=dev(sc), shift=dev(sh), relu=True, row_order=dev(perm))
    plain = ops.spconv_fwd(dev(x), pk, dev(nbr), n, cout, 27, scale=dev(sc), shift=dev(sh), relu=True)
    assert torch.equal(shuf, plain)


@pytest.mark.parametrize("block_bits", [0, 4, 5])
def test_block_index_maps_match_oracle(ops, oracle, block_bits):
    """kernel maps looked up through the block index (bitmap + popcount) == oracle maps, all map kinds, incl. voxels at
    the ends of the 16-bit coordinate range and a fine level probing a coarser one (off-lattice neighbours)."""
    rng = np.random.default_rng(31)
    fine = surface(rng, n=6000, n_batch=3, extent=90)
    edge = np.array([[2, 32767, 32767, 32767], [2, 32766, 32767, 32767], [2, -32768, -32768, -32768], [2, -32767, -32768, -32768]], np.int32)
    fine = np.unique(np.concatenate([fine, edge]), axis=0).astype(np.int32)
    perm = ops.morton_order(dev(fine), 1, block_bits).cpu().numpy()
    fine = fine[perm]
    idx, ndup = ops.block_index_build(dev(fine), 1, block_bits)
    assert ndup == 0 and idx.n_blocks > 3
    same = ops.kernel_map_bi(dev(fine), idx, 3, 1, 1)
    assert np.array_equal(same.cpu().numpy(), oracle.kernel_map(fine, fine, 3, 1, 1))
    assert int(same.pp_pairs) == int((same >= 0).sum())
    mirrored = ops.kernel_map_bi(dev(fine), idx, 3, 1, -1)
    assert np.array_equal(mirrored.cpu().numpy(), oracle.kernel_map(fine, fine, 3, 1, -1))
    for ts in (2, 4):
        coarse = np.unique(np.concatenate([fine[:, :1], fine[:, 1:] // ts * ts], 1), axis=0).astype(np.int32)
        coarse = coarse[ops.morton_order(dev(coarse), ts, block_bits).cpu().numpy()]
        cidx, _ = ops.block_index_build(dev(coarse), ts, block_bits)
        if ts == 2:
            down = ops.kernel_map_bi(dev(coarse), idx, 3, 1, 1)          # coarse rows gather fine rows
            assert np.array_equal(down.cpu().numpy(), oracle.kernel_map(coarse, fine, 3, 1, 1))
            up = ops.kernel_map_bi(dev(fine), cidx, 3, 1, -1)            # fine rows probe the coarse level
            assert np.array_equal(up.cpu().numpy(), oracle.kernel_map(fine, coarse, 3, 1, -1))
            assert torch.equal(up, ops.kernel_map_transpose(down, len(fine)))
        same_c = ops.kernel_map_bi(dev(coarse), cidx, 3, ts, 1)
        assert np.array_equal(same_c.cpu().numpy(), oracle.kernel_map(coarse, coarse, 3, ts, 1))
    # duplicates and unsorted input are reported
    dup = np.concatenate([fine[:10], fine[9:10], fine[10:]])
    assert ops.block_index_build(dev(dup), 1, block_bits)[1] == 1
    with pytest.raises(Exception):
        ops.block_index_build(dev(fine[::-1].copy()), 1, block_bits)


def test_voxelize_and_cylinder_tiles_match_oracle(ops, oracle):
    rng = np.random.default_rng(42)
    pos = (rng.normal(0, 6, size=(60000, 3))).astype(np.float32)
    pos[:500] = np.round(pos[:500] / 0.05) * 0.05 + 0.025
    batch = np.sort(rng.integers(0, 4, size=len(pos)))
    wc, wr, wi = oracle.voxelize(pos, 0.05, batch)
    coords, rep, inv = ops.voxelize(dev(pos), 0.05, dev(batch))
    assert np.array_equal(coords.cpu().numpy(), wc) and np.array_equal(rep.cpu().numpy(), wr)
    assert np.array_equal(inv.cpu().numpy(), wi)
    c2, r2, i2 = ops.voxelize(dev(pos), 0.12)                    # no batch vector
    w2 = oracle.voxelize(pos, 0.12)
    assert np.array_equal(c2.cpu().numpy(), w2[0]) and np.array_equal(r2.cpu().numpy(), w2[1])
    cen = rng.uniform(-8, 8, size=(9, 2)).astype(np.float32)
    csr = ops.cylinder_tiles(dev(pos), dev(cen), 3.0)
    want = oracle.cylinder_tiles(pos, cen, 3.0)
    got = csr.to_list()
    assert len(got) == len(want) == 9
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)
    with pytest.raises(Exception):
        ops.voxelize(dev(np.array([[1e9, 0, 0]], np.float32)), 0.05)


def test_coarser_levels_from_the_block_index(ops, oracle):
    """pp_block_index_coarsen (bit permutation of the occupancy bitmaps) == strided coordinates + sort + index build,
    level after level, incl. several batch elements, voxels at the ends of the range and isolated voxels."""
    rng = np.random.default_rng(51)
    fine = surface(rng, n=9000, n_batch=3, extent=120)
    edge = np.array([[1, 32767, 32767, 32767], [1, 32766, 32766, 32767], [1, -32768, -32768, -32768], [2, -32767, 5, 9]], np.int32)
    fine = np.unique(np.concatenate([fine, edge]), axis=0).astype(np.int32)
    fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
    idx, _ = ops.block_index_build(dev(fine), 1, 4)
    cur_idx, cur = idx, fine
    ts = 1
    for _ in range(5):
        ts *= 2
        got_idx, got = ops.block_index_coarsen(cur_idx, len(cur))
        want = np.unique(np.concatenate([cur[:, :1], cur[:, 1:] // ts * ts], 1), axis=0).astype(np.int32)
        want = want[ops.morton_order(dev(want), ts, 4).cpu().numpy()]
        assert np.array_equal(got.cpu().numpy(), want), ts
        ref_idx, _ = ops.block_index_build(dev(want), ts, 4)
        assert got_idx.n_blocks == ref_idx.n_blocks and got_idx.unit == ts
        nb = ref_idx.n_blocks
        assert torch.equal(got_idx.rec[: nb * 128], ref_idx.rec[: nb * 128])
        assert torch.equal(got_idx.start[:nb], ref_idx.start[:nb]) and torch.equal(got_idx.bkey_ord[:nb], ref_idx.bkey_ord[:nb])
        # maps through the derived index == oracle
        same = ops.kernel_map_bi(got, got_idx, 3, ts, 1)
        assert np.array_equal(same.cpu().numpy(), oracle.kernel_map(want, want, 3, ts, 1))
        down = ops.kernel_map_bi(got, cur_idx, 3, ts // 2, 1)
        assert np.array_equal(down.cpu().numpy(), oracle.kernel_map(want, cur, 3, ts // 2, 1))
        cur_idx, cur = got_idx, want


@pytest.mark.parametrize("levels", [1, 2, 6])
def test_level_chain_equals_level_by_level_coarsening(ops, oracle, levels):
    """pp_block_index_coarsen_chain (all levels of an encoder in one call, the counts of a level feeding the next level's launches
    from device memory, one host read) == `levels` calls of pp_block_index_coarsen: coordinates, records, block starts and keys;
    kernel maps looked up through a chained index == oracle maps.  Several batch elements, range ends, a level chain that
    collapses to one voxel per batch element, and an empty input level."""
    rng = np.random.default_rng(52)
    fine = surface(rng, n=9000, n_batch=3, extent=120)
    edge = np.array([[1, 32767, 32767, 32767], [1, 32766, 32766, 32767], [1, -32768, -32768, -32768], [2, -32767, 5, 9]], np.int32)
    fine = np.unique(np.concatenate([fine, edge]), axis=0).astype(np.int32)
    fine = fine[ops.morton_order(dev(fine), 1, 4).cpu().numpy()]
    idx, _ = ops.block_index_build(dev(fine), 1, 4)
    chain = ops.block_index_coarsen_chain(idx, len(fine), levels)
    assert len(chain) == levels
    cur_idx, cur_n, prev = idx, len(fine), fine
    for l, (got_idx, got) in enumerate(chain):
        ref_idx, ref = ops.block_index_coarsen(cur_idx, cur_n)
        ts = 2 << l
        assert got_idx.unit == ts == ref_idx.unit and got_idx.n_blocks == ref_idx.n_blocks
        assert torch.equal(got, ref), l
        nb = ref_idx.n_blocks
        assert torch.equal(got_idx.rec[: nb * 128], ref_idx.rec[: nb * 128])
        assert torch.equal(got_idx.start[:nb], ref_idx.start[:nb]) and torch.equal(got_idx.bkey_ord[:nb], ref_idx.bkey_ord[:nb])
        c = got.cpu().numpy()
        same = ops.kernel_map_bi(got, got_idx, 3, ts, 1)
        assert np.array_equal(same.cpu().numpy(), oracle.kernel_map(c, c, 3, ts, 1))
        down = ops.kernel_map_bi(got, cur_idx, 3, ts // 2, 1)
        assert np.array_equal(down.cpu().numpy(), oracle.kernel_map(c, prev, 3, ts // 2, 1))
        cur_idx, cur_n, prev = ref_idx, ref.shape[0], c
    # an empty level: every coarser level is empty too
    empty_idx, _ = ops.block_index_build(dev(np.zeros((0, 4), np.int32)), 1, 4)
    for bi, c in ops.block_index_coarsen_chain(empty_idx, 0, 2):
        assert bi.n_blocks == 0 and c.shape == (0, 4)


def test_linear_wgrad_matches_torch(ops):
    """pp_linear_wgrad (streaming dW / db of a skinny Linear layer) vs the float64 product; through modules.Linear's autograd
    path vs torch.nn.Linear; bit-reproducible from run to run"""
    from panopticsegforlargescalepointcloud_amd.modules import Linear
    g = torch.Generator(device="cuda").manual_seed(2)
    for n, cin, cout in [(0, 16, 9), (1, 16, 3), (63, 4, 1), (5000, 16, 16), (300_001, 16, 5), (70_000, 32, 32), (12_345, 7, 13)]:
        x = torch.randn((n, cin), device="cuda", generator=g)
        dy = torch.randn((n, cout), device="cuda", generator=g)
        dw, db = ops.linear_wgrad(x, dy)
        want_w, want_b = (dy.double().T @ x.double()), dy.double().sum(0)
        scale = max(1.0, float(want_w.abs().max())) if n else 1.0
        assert float((dw.double() - want_w).abs().max()) <= 2e-6 * scale * max(1, n) ** 0.5
        assert float((db.double() - want_b).abs().max()) <= 2e-6 * max(1.0, float(want_b.abs().max())) * max(1, n) ** 0.5
        dw2, db2 = ops.linear_wgrad(x, dy)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    torch.manual_seed(0)
    a, b = Linear(16, 9).cuda(), torch.nn.Linear(16, 9).cuda()
    b.load_state_dict(a.state_dict())
    x = torch.randn((20_000, 16), device="cuda", generator=g)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    w = torch.randn((20_000, 9), device="cuda", generator=g)
    (a(xa) * w).sum().backward()
    (b(xb) * w).sum().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-3)
    assert list(a.state_dict()) == list(b.state_dict())


def test_batched_weight_packing_equals_single_layer_packing(ops):
    """ops.pack_weight_cached: after the parameters change (optimizer step) ONE pp_pack_weights_batched launch refreshes every
    registered (layer, orientation); results equal pp_pack_weight's, layer by layer"""
    g = torch.Generator(device="cuda").manual_seed(4)
    cache = ops._PackedWeights()
    params = [torch.nn.Parameter(torch.randn(s, device="cuda", generator=g)) for s in [(27, 16, 16), (27, 32, 48), (16, 32), (27, 4, 16), (27, 64, 32)]]
    combos = [(p, t, k) for p in params for (t, k) in ((False, False), (True, False), (True, True), (False, True)) if not (p.dim() == 2 and k)
              and not (t and p.shape[-1] % 4)]
    first = [cache.get(p, t, k).clone() for p, t, k in combos]
    for (p, t, k), f in zip(combos, first):
        assert torch.equal(f, ops.pack_weight(p, transpose=t, kflip=k))
    with torch.no_grad():
        for p in params:
            p.mul_(1.5).add_(0.25)
    got = [cache.get(p, t, k) for p, t, k in combos]          # first request re-packs all of them in one launch
    for (p, t, k), gt, f in zip(combos, got, first):
        assert torch.equal(gt, ops.pack_weight(p, transpose=t, kflip=k)) and not torch.equal(gt, f)
    tab = cache.tables.get(p.device)
    assert tab is not None and tab[2] == len(combos)


def test_transposed_map_8_wide_equals_dense(ops, oracle):
    """pp_kernel_map_transpose8 / pp_spconv_fwd_t8: the 8-wide form of a stride-2 transposed map holds
    exactly the pairs of the dense 27-wide one (expanded back through the parity classes), its classes are the fine rows'
    coordinate parities, and the convolution on it is bit-identical to the dense form in the same slot order -- fp32 and bf16,
    32 and 64 rows per wave sizes, with and without a slot-ordered strided map behind it"""
    import bruteforce as bf
    rng = np.random.default_rng(12)
    coords = bf.surface_coords(rng, n_batch=3, n=20000, extent=60)
    c = torch.from_numpy(coords).cuda()
    perm, cs = ops.morton_order(c, 1, 4, want_sorted=True, raw=True)
    index, ndup = ops.block_index_build(cs, 1, 4)
    assert ndup == 0
    cidx, cc = ops.block_index_coarsen(index, cs.shape[0])
    n, nc = cs.shape[0], cc.shape[0]
    down = ops.kernel_map_bi(cc, index, 3, 1, 1)                      # coarse rows gather fine rows
    for slot_ordered in (False, True):
        rev, rev_order = down, None
        if slot_ordered:                                                # the strided map in its own slot order
            rev_order = ops.map_order(ops.map_mask(down))
            rev = ops.map_permute(down, rev_order)
        dense = ops.kernel_map_transpose(rev, n, order=rev_order)      # [27, n] physical fine rows
        m8, key = ops.kernel_map_transpose8(rev, n, order=rev_order)
        par = ((cs[:, 1:] & 1) * torch.tensor([1, 2, 4], device="cuda")).sum(1)
        has = (dense >= 0).any(0)
        assert torch.equal((key.long() >> 8)[has], par[has].long())     # class = parity of the fine coordinate (unit 1)
        assert int((m8 >= 0).sum()) == int((dense >= 0).sum())
        assert torch.equal(ops.map8_to_dense(m8), dense)
        order = ops.map_order(key)
        m8s = ops.map_permute(m8, order)
        enc = order
        assert torch.equal(ops.map8_to_dense(m8s), dense[:, order.long()])
        m8s.pp_t8 = True
        dense_s = dense[:, order.long()].contiguous()
        g = torch.Generator(device="cuda").manual_seed(3)
        for cin, cout, bf16 in [(16, 16, False), (64, 64, False), (96, 96, False), (32, 48, True), (160, 160, False)]:
            x = torch.randn((nc, cin), device="cuda", generator=g)
            pk = ops.pack_weight(torch.randn((27, cin, cout), device="cuda", generator=g) * 0.1)
            sc, sh = torch.rand(cout, device="cuda", generator=g) + 0.5, torch.randn(cout, device="cuda", generator=g)
            res = torch.randn((n, cout), device="cuda", generator=g)
            want = ops.spconv_fwd(x, pk, dense_s, n, cout, 27, scale=sc, shift=sh, relu=True, residual=res, row_order=order, bf16=bf16)
            got = ops.spconv_fwd(x, pk, m8s, n, cout, 27, scale=sc, shift=sh, relu=True, residual=res, row_order=enc, bf16=bf16)
            assert torch.equal(got, want), (cin, cout, bf16, slot_ordered)
            # ... and to the register-gather kernel's (k_spconv_x3 instead of k_spconv_x3f where the shape takes the split kernel)
            was = ops.spconv_x3_full_lines(0)
            try:
                frag = ops.spconv_fwd(x, pk, m8s, n, cout, 27, scale=sc, shift=sh, relu=True, residual=res, row_order=enc, bf16=bf16)
            finally:
                ops.spconv_x3_full_lines(was)
            assert torch.equal(got, frag), (cin, cout, bf16, slot_ordered)
    # the oracle's transposed map (fine rows probing the coarse level with mirrored offsets) names the same pairs
    want = oracle.kernel_map(cs.cpu().numpy(), cc.cpu().numpy(), 3, 1, -1)
    assert np.array_equal(ops.map8_to_dense(m8).cpu().numpy(), want)


def test_select_indices_and_run_lengths(ops):
    """pp_select_indices == torch.nonzero, pp_run_lengths == torch.unique_consecutive (+ run boundaries and the run of every
    element), incl. empty input, one run, a run boundary at a scan-tile edge (4096) and no flag set"""
    g = torch.Generator().manual_seed(3)
    for n in (0, 1, 5, 4096, 4097, 300001):
        flags = (torch.rand(n, generator=g) < 0.37).cuda()
        got = ops.select_indices(flags)
        assert torch.equal(got, torch.nonzero(flags).view(-1))
        assert ops.select_indices(torch.zeros(n, dtype=torch.bool).cuda()).numel() == 0
        runs = torch.randint(1, 9000 if n > 10000 else 3, (max(n, 1),), generator=g)
        vals = torch.repeat_interleave(torch.randint(0, 50, (len(runs),), generator=g) * 2 + torch.arange(len(runs)) % 2, runs)[:n].cuda()
        heads, starts, run_id, n_runs = ops.run_lengths(vals)
        uniq, counts = torch.unique_consecutive(vals, return_counts=True)
        nr = int(n_runs.item())
        assert nr == uniq.numel()
        assert torch.equal(heads[:nr], uniq)
        assert torch.equal(starts[: nr + 1].long(), torch.cat([torch.zeros(1, dtype=torch.int64).cuda(), torch.cumsum(counts, 0)]))
        if n:
            assert torch.equal(run_id.long(), torch.repeat_interleave(torch.arange(nr).cuda(), counts))
