"""GPU, BASELINE.json sizes (several million rows per level, the sizes the benchmark's launches have): the oracle cannot follow
there in test time, so the hot path is checked through size-independent properties it must satisfy EXACTLY --
  * kernel maps: the centre offset is the identity, offset k and its mirror 26 - k are inverse relations, the pair count equals
    the number of entries, the transposed map holds the same pairs with the roles swapped, slot order / row translation are
    pure renumberings;
  * convolution: with a one-hot kernel (W[k] = I for one offset, 0 elsewhere) it is the row gather the map names -- bit-exact,
    for every offset, through the 64-rows-per-wave / unsplit variants only launches of this size take; it is linear in the
    features (1e-4); the slot-ordered and the plain map give bit-identical results;
  * coarsening: every fine voxel's parent exists once, rows stay in block order;
  * region growing on a full tile batch (2 M points): clusters are disjoint, uniform in (batch, class), at least
    min_cluster_size large, ascending inside, and identical from run to run (the fixpoint does not depend on the order the
    atomics land in); device sort: sorted, stable, a permutation (30 M pairs).
All through the C ABI (ops -> libpanoptic_hip.so)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N_ROWS = 6_000_000


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from panopticsegforlargescalepointcloud_amd import ops as o
    return o


@pytest.fixture(scope="module")
def level(ops):
    """~6 M unique voxels on noisy surfaces in 24 batch elements, in block order, with index, same-level map and slot order"""
    g = torch.Generator(device="cuda").manual_seed(2022)
    n_raw = int(N_ROWS * 1.25)
    b = torch.randint(0, 24, (n_raw,), device="cuda", generator=g, dtype=torch.int32)
    u = torch.rand((n_raw, 2), device="cuda", generator=g) * 380 - 190
    kind = torch.randint(0, 3, (n_raw,), device="cuda", generator=g)
    h = torch.randn(n_raw, device="cuda", generator=g) * 0.7
    x = torch.where(kind == 1, 40 + h, u[:, 0])                       # ground (z ~ 0), two kinds of walls
    y = torch.where(kind == 2, -25 + h, u[:, 1])
    z = torch.where(kind == 0, h + 0.02 * u[:, 0], (u[:, 0] + u[:, 1]).abs() * 0.2)
    c = torch.stack([b, x.round().int(), y.round().int(), z.round().int()], 1)
    c = torch.unique(c, dim=0)[:N_ROWS].contiguous()
    perm, cs = ops.morton_order(c, 1, 4, want_sorted=True, raw=True)
    index, ndup = ops.block_index_build(cs, 1, 4)
    assert ndup == 0 and cs.shape[0] > 3_000_000
    nbr = ops.kernel_map_bi(cs, index, 3, 1, 1, want_mask=True)
    return {"coords": cs, "index": index, "nbr": nbr, "n": cs.shape[0]}


def test_same_level_map_is_a_symmetric_relation(ops, level):
    nbr, n = level["nbr"], level["n"]
    rows = torch.arange(n, device="cuda", dtype=torch.int32)
    assert torch.equal(nbr[13], rows)                                   # centre offset: every voxel is its own neighbour
    assert int(nbr.pp_pairs) == int((nbr >= 0).sum())
    assert torch.equal(nbr.pp_mask, ops.map_mask(nbr))
    for k in range(13):                                                 # offset k and its mirror are inverse relations
        i = nbr[k]
        has = i >= 0
        back = nbr[26 - k][i[has].long()]
        assert torch.equal(back, rows[has])
        assert int(has.sum()) == int((nbr[26 - k] >= 0).sum())
    assert torch.equal(ops.kernel_map_bi(level["coords"], level["index"], 3, 1, -1), torch.flip(nbr, [0]))


def test_slot_order_is_a_renumbering_and_conv_does_not_see_it(ops, level):
    nbr, n, coords = level["nbr"], level["n"], level["coords"]
    order = ops.map_order(nbr.pp_mask)
    w = order.pp_window
    o = order.long()
    assert torch.equal(torch.sort(o)[0], torch.arange(n, device="cuda"))
    assert torch.equal(o // w, torch.arange(n, device="cuda") // w)       # rows never leave their window
    coords_p, phys_of = ops.level_permute(coords, order)
    assert torch.equal(coords_p, coords[o]) and torch.equal(phys_of[o].long(), torch.arange(n, device="cuda"))
    same = ops.map_permute(nbr, order, translate=phys_of)
    # same relation, renumbered: same[k][s] = phys_of[nbr[k][order[s]]]
    for k in (0, 4, 13, 22, 26):
        want = nbr[k][o]
        want = torch.where(want >= 0, phys_of[want.clamp(min=0).long()], want)
        assert torch.equal(same[k], want)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((n, 16), device="cuda", generator=g)
    wgt = torch.randn((27, 16, 16), device="cuda", generator=g) * 0.1
    pk = ops.pack_weight(wgt)
    plain = ops.spconv_fwd(x, pk, nbr, n, 16, 27)
    slot = ops.spconv_fwd(x[o], pk, same, n, 16, 27)
    assert torch.equal(slot, plain[o])                                   # bit-identical: per-row summation order is the same
    level["order"], level["same"], level["phys_of"] = order, same, phys_of


@pytest.mark.parametrize("cin,cout", [(16, 16), (4, 16), (32, 32), (64, 48)])
def test_one_hot_kernel_is_the_gather_the_map_names(ops, level, cin, cout):
    """exact at any size: with W[k] = [I | 0] the convolution must return the input row nbr[k][o] (zeros where there is none)"""
    nbr, n = level["nbr"], level["n"]
    g = torch.Generator(device="cuda").manual_seed(cin)
    x = torch.randn((n, cin), device="cuda", generator=g)
    c = min(cin, cout)
    for k in (0, 13, 17, 26):
        wgt = torch.zeros((27, cin, cout), device="cuda")
        wgt[k, :c, :c] = torch.eye(c, device="cuda")
        out = ops.spconv_fwd(x, ops.pack_weight(wgt), nbr, n, cout, 27)
        i = nbr[k]
        want = torch.zeros((n, cout), device="cuda")
        want[:, :c] = torch.where((i >= 0)[:, None], x[i.clamp(min=0).long()][:, :c], torch.zeros((), device="cuda"))
        assert torch.equal(out, want), (cin, cout, k)


def test_convolution_is_linear_at_full_size(ops, level):
    nbr, n = level["nbr"], level["n"]
    g = torch.Generator(device="cuda").manual_seed(9)
    x, y = torch.randn((n, 32), device="cuda", generator=g), torch.randn((n, 32), device="cuda", generator=g)
    pk = ops.pack_weight(torch.randn((27, 32, 32), device="cuda", generator=g) * 0.05)
    fx, fy = ops.spconv_fwd(x, pk, nbr, n, 32, 27), ops.spconv_fwd(y, pk, nbr, n, 32, 27)
    fz = ops.spconv_fwd(2.0 * x - 0.5 * y, pk, nbr, n, 32, 27)
    torch.testing.assert_close(fz, 2.0 * fx - 0.5 * fy, rtol=1e-4, atol=5e-5)   # fp32 rounding of ~200-term sums of O(1) values
    # fused epilogue at this size: BN scale / shift, ReLU, residual
    sc, sh = torch.rand(32, device="cuda", generator=g) + 0.5, torch.randn(32, device="cuda", generator=g)
    full = ops.spconv_fwd(x, pk, nbr, n, 32, 27, scale=sc, shift=sh, relu=True, residual=y)
    torch.testing.assert_close(full, torch.relu(fx * sc + sh) + y, rtol=1e-5, atol=1e-5)         # same sums, epilogue only


def test_coarse_level_and_cross_level_maps(ops, level):
    coords, index, n = level["coords"], level["index"], level["n"]
    cidx, cc = ops.block_index_coarsen(index, n)
    nc = cc.shape[0]
    parents = torch.cat([coords[:, :1], coords[:, 1:] // 2 * 2], 1)
    assert nc == torch.unique(parents, dim=0).shape[0]                   # every occupied parent exactly once
    down = ops.kernel_map_bi(cc, index, 3, 1, 1)                         # coarse rows gather fine rows
    assert int(down.pp_pairs) == int((down >= 0).sum())
    # a fine voxel at offset d from its parent's corner is found from the parent through offsets {0,1}^3 only
    fine_of = down[13]
    has = fine_of >= 0
    assert torch.equal(coords[fine_of[has].long()], cc[has])
    up = ops.kernel_map_transpose(down, n)
    assert int((up >= 0).sum()) == int(down.pp_pairs)
    for k in (0, 13, 26):                                                # up[k][i] = c  <=>  down[k][c] = i
        c_of = up[k]
        h = c_of >= 0
        assert torch.equal(down[k][c_of[h].long()].long(), torch.nonzero(h).view(-1))
    # every fine voxel reaches its parent through exactly one of the 8 non-negative offsets
    reach = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k in range(27):
        dx, dy, dz = k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1
        if min(dx, dy, dz) >= 0:
            c_of = up[k]
            h = c_of >= 0
            is_parent = torch.zeros(n, dtype=torch.bool, device="cuda")
            is_parent[h] = (cc[c_of[h].long()] == parents[h]).all(1)
            reach += is_parent.int()
    assert bool((reach == 1).all())


def test_sort_is_sorted_stable_and_a_permutation(ops):
    n = 30_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    keys = torch.randint(0, 2 ** 40, (n,), dtype=torch.int64, device="cuda", generator=g)
    keys[::3] = keys[1::3][: keys[::3].shape[0]]
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo = ops.sort_pairs(keys, vals, 40)
    assert bool((ko[1:] >= ko[:-1]).all())                               # sorted
    assert bool(((ko[1:] > ko[:-1]) | (vo[1:] > vo[:-1])).all())         # stable: ties keep ascending input positions
    assert torch.equal(keys[vo.long()], ko)                              # the values name the rows the keys came from
    seen = torch.zeros(n, dtype=torch.bool, device="cuda")
    seen[vo.long()] = True
    assert bool(seen.all())                                              # a permutation


def test_region_growing_properties_on_a_full_batch(ops):
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    import bench
    scene, tiles, _ = bench.build_scene(2_000_000, 4, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(len(tiles))))
    cls, off, _ = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(1))
    pos = torch.from_numpy(b["pos"] + off).cuda()
    pred, batch = torch.from_numpy(cls).cuda(), torch.from_numpy(b["batch"]).cuda()
    ignore = torch.tensor(syn.NPM3D_STUFF)
    radius = 0.075
    csr, pc = ops.region_grow_csr(pos, pred, batch, ignore, 200, radius, 10, syn.NPM3D_NUM_CLASSES)
    assert csr.n > 100
    pts, offs = csr.points, csr.offsets.long()
    sizes = offs[1:] - offs[:-1]
    assert int(sizes.min()) >= 10 and int(sizes.sum()) == pts.shape[0]
    assert torch.unique(pts).shape[0] == pts.shape[0]                    # clusters are disjoint
    cid = torch.repeat_interleave(torch.arange(csr.n, device="cuda"), sizes)
    assert torch.equal(pc[pts], cid.int()) and int((pc >= 0).sum()) == pts.shape[0]
    first = pts[offs[:-1]]
    assert torch.equal(pred[pts], pred[first][cid]) and torch.equal(batch[pts], batch[first][cid])   # uniform (batch, class)
    assert not bool(torch.isin(pred[pts], ignore.cuda()).any())
    inside = torch.ones_like(pts, dtype=torch.bool)
    inside[offs[:-1]] = False
    assert bool((pts[1:] > pts[:-1])[inside[1:]].all())                  # points ascend inside a cluster
    # deterministic: the fixpoint does not depend on the order the atomics land in
    csr2, pc2 = ops.region_grow_csr(pos, pred, batch, ignore, 200, radius, 10, syn.NPM3D_NUM_CLASSES)
    assert torch.equal(pc2, pc) and torch.equal(csr2.points, pts) and torch.equal(csr2.offsets, csr.offsets)


@pytest.mark.parametrize("c,ds_c,bf16", [(16, 64, False), (32, 96, False), (64, 32, False), (48, 128, False), (32, 64, True)])
def test_fused_shortcut_equals_two_launches(ops, level, c, ds_c, bf16):
    """the 1x1 shortcut of a residual block fused into the block's last convolution (pp_spconv_fwd_shortcut) is bit-identical
    to the separate 1x1 convolution + residual add it replaces; small launches (split-K) are not served (None, nothing run)"""
    nbr, n = level["nbr"], level["n"]
    g = torch.Generator(device="cuda").manual_seed(c + ds_c)
    h = torch.randn((n, c), device="cuda", generator=g)
    x = torch.randn((n, ds_c), device="cuda", generator=g)
    pk = ops.pack_weight(torch.randn((27, c, c), device="cuda", generator=g) * 0.05)
    pk1 = ops.pack_weight(torch.randn((ds_c, c), device="cuda", generator=g) * 0.1)
    sc, sh = torch.rand(c, device="cuda", generator=g) + 0.5, torch.randn(c, device="cuda", generator=g)
    sc1, sh1 = torch.rand(c, device="cuda", generator=g) + 0.5, torch.randn(c, device="cuda", generator=g)
    res = ops.spconv_fwd(x, pk1, None, n, c, 1, scale=sc1, shift=sh1, relu=False, bf16=bf16)
    want = ops.spconv_fwd(h, pk, nbr, n, c, 27, scale=sc, shift=sh, relu=True, residual=res, bf16=bf16)
    got = ops.spconv_fwd(h, pk, nbr, n, c, 27, scale=sc, shift=sh, relu=True, bf16=bf16, shortcut=(x, pk1, sc1, sh1))
    assert got is not None and torch.equal(got, want)
    m = 3000                                                            # a launch this small is split over the offsets
    small = ops.spconv_fwd(h[:m].contiguous(), pk, nbr[:, :m].clamp(max=m - 1).contiguous(), m, c, 27, scale=sc, shift=sh, relu=True,
                           shortcut=(x[:m].contiguous(), pk1, sc1, sh1))
    assert small is None


# ---------------------------------------------------------------------------------------------------------------------
# numeric parity at full size: general weights, fp64 evaluation of SAMPLED output rows from the same map
# (api_modules.py:9-82: convolution + BatchNorm + ReLU (+ residual, + cat) as ONE launch; north_star bar 1e-4)
# ---------------------------------------------------------------------------------------------------------------------
def _rows_fp64(x0, w, nbr, rows, x1=None, scale=None, shift=None, relu=False, residual=None, out_rows=None, bf16=False):
    """out[rows] = epilogue(sum_k concat(x0, x1)[nbr[k][slot]] @ w[k]) in float64.  rows index the map's columns (slots);
    out_rows = the physical output rows those slots write (residual rows).  bf16: operands rounded as the kernel rounds."""
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    if bf16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    idx = nbr[:, rows].long()                                            # [27, S]
    g = x[idx.clamp(min=0)].double() * (idx >= 0)[..., None]             # [27, S, Cin]
    out = torch.einsum("ksc,kcd->sd", g, w.double())
    if scale is not None:
        out = out * scale.double() + shift.double()
    if relu:
        out = out.clamp(min=0)
    if residual is not None:
        out = out + residual[rows if out_rows is None else out_rows].double()
    return out


def _report(name, got, want):
    err = (got.double() - want).abs()
    scale = max(1.0, float(want.abs().max()))
    mx, rel = float(err.max()), float((err / want.abs().clamp(min=1e-3)).max())
    print("fullsize parity %-34s max abs err %.3e  (scale %.2f -> %.3e of it)  max rel err %.3e" % (name, mx, scale, mx / scale, rel))
    return mx / scale


@pytest.mark.parametrize("cin,cout,cat,bf16", [(16, 16, False, False), (32, 32, False, False), (64, 48, False, False),
                                               (48, 48, True, False), (4, 16, False, False), (32, 32, False, True),
                                               (64, 64, True, True), (32, 96, False, False), (48, 80, False, False),
                                               (16, 160, False, True)])
def test_general_weight_convolution_vs_fp64_on_sampled_rows(ops, level, cin, cout, cat, bf16):
    """same-level convolution with the fused epilogue (cat + BN + ReLU + residual) at the size where the 64-rows-per-wave /
    unsplit variants are selected -- and, for 80 / 96 / 160 output channels, the 5 / 6-column-tiles-per-wave variants only
    launches of >= 0.4 M rows take -- against a float64 evaluation of 20 k sampled output rows; 1e-4 of the output magnitude"""
    nbr, n = level["nbr"], level["n"]
    g = torch.Generator(device="cuda").manual_seed(100 + cin + cout)
    x0 = torch.randn((n, cin), device="cuda", generator=g)
    x1 = torch.randn((n, cin), device="cuda", generator=g) if cat else None
    w = torch.randn((27, cin * (2 if cat else 1), cout), device="cuda", generator=g) * (0.5 / np.sqrt(cin * (2 if cat else 1)))
    sc, sh = torch.rand(cout, device="cuda", generator=g) + 0.5, torch.randn(cout, device="cuda", generator=g) * 0.3
    res = torch.randn((n, cout), device="cuda", generator=g)
    rows = torch.randint(0, n, (20_000,), device="cuda", generator=g)
    got = ops.spconv_fwd(x0, ops.pack_weight(w), nbr, n, cout, 27, in1=x1, scale=sc, shift=sh, relu=True, residual=res, bf16=bf16)
    want = _rows_fp64(x0, w, nbr, rows, x1=x1, scale=sc, shift=sh, relu=True, residual=res, bf16=bf16)
    assert _report("same %d%s->%d%s" % (cin, "+%d" % cin if cat else "", cout, " bf16" if bf16 else ""), got[rows], want) < 1e-4


@pytest.mark.parametrize("c,bf16", [(32, False), (64, False), (64, True), (96, False)])
def test_strided_and_transposed_convolutions_vs_fp64_on_sampled_rows(ops, level, c, bf16):
    """strided (fine -> coarse) and transposed (coarse -> fine, slot-ordered map with row_order) launches of this size"""
    coords, index, n = level["coords"], level["index"], level["n"]
    cidx, cc = ops.block_index_coarsen(index, n)
    nc = cc.shape[0]
    g = torch.Generator(device="cuda").manual_seed(7 + c)
    w = torch.randn((27, c, c), device="cuda", generator=g) * (0.5 / np.sqrt(c))
    xf, xc = torch.randn((n, c), device="cuda", generator=g), torch.randn((nc, c), device="cuda", generator=g)
    down = ops.kernel_map_bi(cc, index, 3, 1, 1)
    rows = torch.randint(0, nc, (20_000,), device="cuda", generator=g)
    got = ops.spconv_fwd(xf, ops.pack_weight(w), down, nc, c, 27, relu=True, bf16=bf16)
    assert _report("strided %d->%d%s" % (c, c, " bf16" if bf16 else ""), got[rows],
                   _rows_fp64(xf, w, down, rows, relu=True, bf16=bf16)) < 1e-4
    up = ops.kernel_map_transpose(down, n)                                # physical fine rows
    order = ops.map_order(ops.map_mask(up))                               # its own slot order, as the coordinate manager builds it
    up_s = ops.map_permute(up, order)
    slots = torch.randint(0, n, (20_000,), device="cuda", generator=g)
    got = ops.spconv_fwd(xc, ops.pack_weight(w), up_s, n, c, 27, row_order=order, bf16=bf16)
    want = _rows_fp64(xc, w, up_s, slots, bf16=bf16)
    assert _report("transposed %d->%d%s" % (c, c, " bf16" if bf16 else ""), got[order[slots].long()], want) < 1e-4


@pytest.mark.parametrize("cin,cout,bf16", [(16, 16, False), (32, 48, False), (64, 64, False), (32, 32, True)])
def test_weight_gradient_at_full_size(ops, level, cin, cout, bf16):
    """dW over the ~6 M-row level (the 32-bit list positions, the chunk grid and the block reduction at the size of a real
    launch; the deterministic form without its fall-back): the pair lists hold exactly the map's pairs per offset, in row order; dW from the lists equals dW from the dense
    map (two independent kernels, different summation orders: 1e-4 of the largest entry) and, for three offsets, an fp64
    evaluation of the definition (torch index arithmetic); a slot-ordered copy of the map gives the same dW from the same
    output gradient."""
    nbr, n = level["nbr"], level["n"]
    g = torch.Generator(device="cuda").manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, device="cuda", generator=g)
    dy = torch.randn(n, cout, device="cuda", generator=g)
    wp = ops.wgrad_pairs(nbr, 27)
    T = (n + 1023) // 1024
    ts = wp.tile_start.long()
    per_k = ts[torch.arange(1, 28, device="cuda") * T] - ts[torch.arange(0, 27, device="cuda") * T]
    assert torch.equal(per_k, (nbr >= 0).sum(1)) and int(ts[-1]) == int(nbr.pp_pairs)
    for k in (0, 13, 26):
        lo, hi = int(ts[k * T]), int(ts[(k + 1) * T])
        rows = torch.nonzero(nbr[k] >= 0).view(-1)
        assert torch.equal(wp.pairs[lo:hi, 0].long(), rows) and torch.equal(wp.pairs[lo:hi, 1], nbr[k][rows])
    dw = ops.spconv_bwd_weight_pairs(x, dy, wp, bf16=bf16)
    # the ordered reduction serves this size too (round 6: a wave walks more pairs when the block partials would exceed
    # PP_WGRAD_DET_MAX_MB; the float-atomic fall-back of round 5 is gone): the same bits run after run, no warning
    assert torch.equal(dw, ops.spconv_bwd_weight_pairs(x, dy, wp, bf16=bf16))
    assert not ops._WGRAD_WARNED[0], "the deterministic weight gradient fell back to float atomics"
    dense = ops.spconv_bwd_weight(x, dy, nbr, 27, bf16=bf16)
    scale = float(dense.abs().max())
    err = float((dw - dense).abs().max()) / scale
    print("dW %d->%d%s at %d rows: pair lists vs dense map %.2e of the largest entry" % (cin, cout, " bf16" if bf16 else "", n, err))
    assert err < 1e-4
    xr, dr = (x.bfloat16().double(), dy.bfloat16().double()) if bf16 else (x.double(), dy.double())
    for k in (1, 13, 22):
        rows = torch.nonzero(nbr[k] >= 0).view(-1)
        want = xr[nbr[k][rows].long()].t() @ dr[rows]
        e = float((dw[k].double() - want).abs().max()) / float(want.abs().max())
        print("   offset %d vs fp64: %.2e" % (k, e))
        assert e < 1e-4
    order = ops.map_order(nbr.pp_mask)[:n]
    slot = ops.map_permute(nbr, order)
    wps = ops.wgrad_pairs(slot, 27, row_order=order)
    dws = ops.spconv_bwd_weight_pairs(x, dy, wps, bf16=bf16)
    assert float((dws - dense).abs().max()) / scale < 1e-4


def test_region_growing_tile_inside_full_batch_equals_oracle(ops, oracle):
    """one tile of the 2 M-point batch through the literal sequential oracle: the batched launch must give that tile the
    same clusters (bit-exact) as the CPU restatement run on the tile alone"""
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    import bench
    scene, tiles, _ = bench.build_scene(2_000_000, 4, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(len(tiles))))
    cls, off, _ = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(1))
    shifted = (b["pos"] + off).astype(np.float32)
    pos, pred, batch = torch.from_numpy(shifted).cuda(), torch.from_numpy(cls).cuda(), torch.from_numpy(b["batch"]).cuda()
    ignore = torch.tensor(syn.NPM3D_STUFF)
    csr, pc = ops.region_grow_csr(pos, pred, batch, ignore, 200, 0.075, 10, syn.NPM3D_NUM_CLASSES)
    sizes = np.bincount(b["batch"])
    t = int(np.argsort(sizes)[len(sizes) // 2])                           # the median tile
    m = b["batch"] == t
    local = np.nonzero(m)[0]
    want, _ = oracle.region_grow(shifted[m], cls[m].astype(np.int64), np.zeros(int(m.sum()), np.int64),
                                 ignore_labels=syn.NPM3D_STUFF, nsample=200, radius=0.075, min_cluster_size=10)
    got = [c.cpu().numpy() for c in csr.to_list()]
    got_t = [c for c in got if m[c[0]]]
    assert len(got_t) == len(want) and len(want) > 3
    for gc, wc in zip(got_t, want):
        assert np.array_equal(gc, local[np.sort(np.asarray(wc))])
    print("fullsize parity region growing: tile %d (%d points, %d clusters) inside a %d-point batch = oracle" % (
        t, int(m.sum()), len(want), len(m)))
