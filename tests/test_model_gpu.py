"""GPU: the whole hot path (sparse U-Net -> heads -> grouping -> ScorerUnet -> NMS labels) through the product
modules vs the CPU oracle pipeline on the same seeded inputs and weights."""
import os

import numpy as np
import pytest
import torch

import bruteforce as bf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def setup():
    import bench
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner
    scene, tiles, radius = bench.build_scene(60_000, 2, 0.05, 2022)
    model, cfg, DS = bench.build_model(torch.device("cuda"), 0.05)
    ids = [0, 3]
    b = syn.tile_batch(scene, tiles, ids)
    rng = np.random.default_rng(5)
    cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, rng)
    return dict(scene=scene, tiles=tiles, model=model, cfg=cfg, DS=DS, b=b, ids=ids, override=(cls, off, emb),
                runner=TileRunner(model, torch.device("cuda")))


_err = bf.scaled_err


def _oracle_forward(s, override):
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    sd = {k: v.detach().cpu() for k, v in s["model"].state_dict().items()}
    opt = {"cluster_radius_search": s["cfg"].cluster_radius_search, "cluster_type": s["cfg"].cluster_type,
           "bandwidth": s["cfg"].bandwidth}
    out = opipe.forward(sd, s["b"], opt, 9, syn.NPM3D_STUFF, override=override)
    labels = opipe.instance_labels(out, len(s["b"]["pos"]), s["b"]["batch"])
    return out, labels


def test_full_path_matches_oracle_with_synthetic_head_statistics(setup):
    s = setup
    dev = torch.device("cuda")
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])
    labels, res, counts = s["runner"].run(s["b"], len(s["ids"]), override=ov)
    want, want_labels = _oracle_forward(s, s["override"])
    feats = s["model"].Backbone(s["model"].input).x if False else None  # features are checked through the heads below
    # network outputs (not overridden): within 1e-4 float32
    assert _err("semantic log-probs", res.semantic_logits.cpu().numpy(), want["semantic_logits"]) < 1e-4
    # grouping on identical inputs: bit-exact proposals (same order: region growing first, then mean shift)
    got = [c.cpu().numpy() for c in res.clusters_csr.to_list()]
    assert len(got) == len(want["clusters"]) and len(got) > 4
    for g, w in zip(got, want["clusters"]):
        assert np.array_equal(g, np.sort(w))
    assert np.array_equal(res.cluster_type.cpu().numpy(), want["cluster_type"])
    assert _err("proposal scores", res.cluster_scores.cpu().numpy(), want["cluster_scores"]) < 1e-4
    # NMS + painting: bit-exact after label canonicalisation GIVEN THE SAME SCORES (a random-init scorer squeezes all
    # scores into a ~1e-3 band, so float-rounding differences between the two score vectors may legitimately swap the
    # paint order of two overlapping proposals; the scores themselves are compared above)
    from oracle import pipeline as opipe
    want2 = dict(want)
    want2["cluster_scores"] = res.cluster_scores.cpu().numpy()
    want_labels = opipe.instance_labels(want2, len(s["b"]["pos"]), s["b"]["batch"])
    b = s["b"]["batch"]
    for t in range(len(s["ids"])):
        m = b == t
        assert np.array_equal(bf.canon_partition(labels.cpu().numpy()[m]), bf.canon_partition(want_labels[m]))
    assert sum(counts) > 0


def test_instance_labels_match_oracle_without_score_substitution(setup):
    """the ScorerHead of a random-init model squeezes all scores into a ~1e-3 band, where float rounding decides the paint
    order; here its Linear layer is rescaled (same features, logits spread to roughly [0.2, 4] => scores over (0.55, 0.98)) so
    that the scores are SEPARATED, and the GPU's instance labels are compared with the oracle's own -- oracle scores, oracle
    NMS, no substitution"""
    s = setup
    dev = torch.device("cuda")
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])
    _, res0, _ = s["runner"].run(s["b"], len(s["ids"]), override=ov)
    with bf.spread_scorer_head(s["model"].ScorerHead[0], res0.cluster_scores):
        labels, res, counts = s["runner"].run(s["b"], len(s["ids"]), override=ov)
        want, want_labels = _oracle_forward(s, s["override"])
    got_sc, want_sc = res.cluster_scores.cpu().numpy(), want["cluster_scores"]
    gaps = np.diff(np.sort(want_sc))
    print("score spread: min %.3f max %.3f, median gap between neighbours %.2e, smallest %.2e" % (
        want_sc.min(), want_sc.max(), np.median(gaps), gaps.min()))
    assert want_sc.max() - want_sc.min() > 0.2
    assert _err("proposal scores (spread)", got_sc, want_sc) < 1e-4
    b = s["b"]["batch"]
    for t in range(len(s["ids"])):
        m = b == t
        assert np.array_equal(bf.canon_partition(labels.cpu().numpy()[m]), bf.canon_partition(want_labels[m]))
    assert sum(counts) > 0


@pytest.mark.parametrize("min_rows", [2, 8000])
def test_network_outputs_match_oracle(setup, monkeypatch, min_rows):
    """min_rows = MAP_ORDER_MIN_ROWS: 2 slot-orders every level (the conftest default for the small test clouds); 8000 is the
    production situation -- the fine levels are slot-ordered, the coarse ones keep the block order, so ordered and un-ordered
    levels meet in the strided / transposed maps (translate / phys_of on one side only)"""
    s = setup
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    from panopticsegforlargescalepointcloud_amd.applications import Data
    monkeypatch.setattr(ME, "MAP_ORDER_MIN_ROWS", min_rows)
    dev = torch.device("cuda")
    m = s["model"]
    data = Data(**{k: torch.from_numpy(v).to(dev) for k, v in s["b"].items()})
    m.set_input(data, dev)
    with torch.no_grad():
        feats, sem, off, emb, pred = m.backbone_and_heads()
    want, _ = _oracle_forward(s, None)
    # north_star: embeddings within 1e-4 fp32 -- asserted for every float output, as a fraction of its magnitude
    assert _err("backbone features", feats.cpu().numpy(), want["features"]) < 1e-4
    assert _err("offsets", off.cpu().numpy(), want["offset_logits"]) < 1e-4
    assert _err("embeddings", emb.cpu().numpy(), want["embed_logits"]) < 1e-4
    assert _err("semantic log-probs", sem.cpu().numpy(), want["semantic_logits"]) < 1e-4
    assert (pred.cpu().numpy() != want["pred"]).mean() < 1e-3
    # row order: output row i belongs to input row i (applications/minkowski.py:193)
    assert feats.shape[0] == len(s["b"]["pos"])


def test_reference_list_api_and_get_instances(setup):
    s = setup
    from panopticsegforlargescalepointcloud_amd.applications import Data
    dev = torch.device("cuda")
    m = s["model"]
    m.set_input(Data(**{k: torch.from_numpy(v).to(dev) for k, v in s["b"].items()}), dev)
    with torch.no_grad():
        out = m.forward(epoch=100)
    assert isinstance(out.clusters, list) and len(out.clusters) == out.clusters_csr.n
    ids, clusters = out.get_instances(min_cluster_points=10)
    assert len(ids) == len(clusters)
    out0 = m.forward(epoch=1)  # before prepare_epoch: no grouping (PointGroup3heads.py:115-116)
    assert out0.clusters is None and out0.cluster_scores is None


@pytest.mark.parametrize("min_rows", [2, 1200])
def test_training_step_runs_and_matches_torch_autograd(setup, monkeypatch, min_rows):
    """fwd + bwd through the unfused autograd path: sparse conv gradients vs a dense torch re-implementation.
    min_rows = 1200: the fine level is slot-ordered, the strided level is not (the production mix)."""
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    monkeypatch.setattr(ME, "MAP_ORDER_MIN_ROWS", min_rows)
    rng = np.random.default_rng(3)
    coords = bf.surface_coords(rng, n_batch=2, n=1500, extent=30)
    c = torch.from_numpy(coords).cuda()
    x = torch.randn(len(coords), 16, device="cuda", requires_grad=True)
    torch.manual_seed(1)
    conv = ME.MinkowskiConvolution(16, 32, kernel_size=3, stride=2, dimension=3).cuda()
    bn = ME.MinkowskiBatchNorm(32).cuda()
    up = ME.MinkowskiConvolutionTranspose(32, 16, kernel_size=3, stride=2, dimension=3).cuda()
    st = ME.SparseTensor(features=x, coordinates=c, device="cuda")
    y = up(ME.MinkowskiReLU()(bn(conv(st))))
    loss = (y.F ** 2).mean()
    loss.backward()
    assert int(bn.bn.num_batches_tracked) == 1   # nn.BatchNorm's counter, incremented on the device by the stats launch
    # dense re-implementation with index_add on the same kernel maps
    cm = st.coordinate_manager
    down = cm.kernel_map_rows(1, 2, 3, 1).long()   # indexed by physical output rows (the conv kernels use slot order)
    upm = cm.kernel_map_rows(2, 1, 3, -1).long()
    x2 = x.detach().clone().requires_grad_(True)
    w1 = conv.kernel.detach().clone().requires_grad_(True)
    w2 = up.kernel.detach().clone().requires_grad_(True)
    n2 = cm.level(2).n

    def gconv(inp, w, nbr, n_out):
        out = torch.zeros(n_out, w.shape[2], device="cuda")
        for k in range(27):
            r = nbr[k]
            ok = r >= 0
            out = out.index_add(0, torch.nonzero(ok).view(-1), inp[r[ok]] @ w[k])
        return out

    xin = x2 if cm.perm is None else x2[cm.perm]  # kernel maps index the manager's internal (Morton) row order
    h = gconv(xin, w1, down, n2)
    h = torch.nn.functional.batch_norm(h, None, None, bn.bn.weight.detach(), bn.bn.bias.detach(), True, 0.1, 1e-5)
    h = torch.relu(h)
    y2 = gconv(h, w2, upm, len(coords))
    loss2 = (y2 ** 2).mean()
    loss2.backward()
    np.testing.assert_allclose(float(loss), float(loss2), rtol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(conv.kernel.grad.cpu().numpy(), w1.grad.cpu().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(up.kernel.grad.cpu().numpy(), w2.grad.cpu().numpy(), rtol=1e-3, atol=1e-5)


def test_fused_training_node_is_bit_identical_to_module_by_module(monkeypatch):
    """ME.conv_bn_act_train (one autograd node per conv + BN + ReLU) against the module-by-module path: same launches in the
    same order, so outputs, running statistics and every gradient except the atomically accumulated kernel gradients agree bit
    for bit (those to float rounding); a residual block with a 1x1 shortcut, a strided and a transposed input convolution."""
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, modules as M
    rng = np.random.default_rng(9)
    coords = torch.from_numpy(bf.surface_coords(rng, n_batch=2, n=2500, extent=36)).cuda()
    x0 = torch.randn(len(coords), 16, device="cuda")
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(ME, "FUSE_TRAIN", fused)
        torch.manual_seed(4)
        down = M.ResNetDown(down_conv_nn=[16, 32], kernel_size=3, stride=2, N=1).cuda().train()
        up = M.ResNetUp(up_conv_nn=[32 + 32, 16], kernel_size=3, stride=2, N=1).cuda().train()
        x = x0.clone().requires_grad_(True)
        st = ME.SparseTensor(features=x, coordinates=coords, device="cuda")
        h = down(st)
        y = up(h, h)          # (skip and input on the same level, as in the U-Net)
        (y.F ** 2).mean().backward()
        outs[fused] = (y.F.detach(), x.grad, [(n, p.grad) for n, p in list(down.named_parameters()) + list(up.named_parameters())],
                       [(n, b.clone()) for n, b in list(down.named_buffers()) + list(up.named_buffers())])
    ya, xa, pa, ba = outs[True]
    yb, xb, pb, bb = outs[False]
    assert torch.equal(ya, yb) and torch.equal(xa, xb)
    for (n, a), (_, b) in zip(ba, bb):
        assert torch.equal(a, b), n
    assert any(n.endswith("num_batches_tracked") and int(v) == 1 for n, v in ba)
    for (n, a), (_, b) in zip(pa, pb):
        if n.endswith("kernel"):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
        else:
            assert torch.equal(a, b), n


_LOSSES_BY_OPT = {}


@pytest.mark.parametrize("fused", [False, True])
def test_full_model_training_step(setup, fused):
    """optimize_parameters2 semantics (models/base_model.py:259-285): forward in train mode (batch-statistics BN through
    the HIP stats kernels), _compute_loss, backward through every sparse conv, Adam step; loss finite and decreasing.
    fused: torch's fused Adam writes the parameters without bumping their version counters -- the packed-weight cache must
    follow the optimizer step all the same (the losses of the two optimizers agree step by step)."""
    import bench
    from panopticsegforlargescalepointcloud_amd.applications import Data
    s = setup
    dev = torch.device("cuda")
    model = bench.build_model(dev, 0.05)[0].train()
    scene, b = s["scene"], s["b"]
    oid = b["origin_id"]
    inst = scene.inst[oid]
    # labels in the reference layout (datasets/panoptic/utils.py:41-48): ids 1..k per batch element, 0 = none
    inst_local = np.zeros_like(inst)
    for t in np.unique(b["batch"]):
        m = (b["batch"] == t) & (inst > 0)
        _, inv = np.unique(inst[m], return_inverse=True)
        inst_local[m] = inv + 1
    vote = (scene.inst_center[inst] - scene.pos[oid]).astype(np.float32)
    data = Data(pos=torch.from_numpy(b["pos"]), coords=torch.from_numpy(b["coords"]), x=torch.from_numpy(b["x"]),
                batch=torch.from_numpy(b["batch"]), y=torch.from_numpy(scene.cls[oid]),
                instance_labels=torch.from_numpy(inst_local), instance_mask=torch.from_numpy(inst > 0),
                vote_label=torch.from_numpy(vote), center_label=torch.from_numpy(scene.inst_center[inst]),
                num_instances=torch.tensor([int(inst_local.max())]))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=fused)
    losses = []
    for it in range(3):
        model.set_input(data, dev)
        opt.zero_grad()
        model.forward(epoch=1)  # epoch <= prepare_epoch: heads only, as in the reference's first 30 epochs
        model.backward(1)
        opt.step()
        losses.append(float(model.loss))
        assert np.isfinite(losses[-1])
    g = [p.grad for n, p in model.named_parameters() if n.startswith("Backbone") and p.grad is not None]
    assert len(g) > 150 and all(torch.isfinite(x).all() for x in g)
    assert losses[-1] < losses[0]
    _LOSSES_BY_OPT[fused] = losses
    if len(_LOSSES_BY_OPT) == 2:
        np.testing.assert_allclose(_LOSSES_BY_OPT[True], _LOSSES_BY_OPT[False], rtol=2e-3)
    cur = model.get_current_losses()
    assert set(("loss", "semantic_loss", "offset_norm_loss", "ins_loss")) <= set(cur)


def test_training_gradients_run_to_run(setup):
    """Two forward + backward passes from the SAME model state and batch give the SAME BITS: the convolutions' weight
    gradients are block partials added in a fixed order (pp_spconv_bwd_weight_pairs_det), BatchNorm statistics are float64
    block partials, every convolution row is summed in a fixed order, and the segment sums of the losses and of the row
    gathers' backward go through pp_segment_sum_ordered (rows grouped by segment, one workgroup per segment adding in an order
    fixed by the segment's size) instead of float atomics.  With PP_SEGMENT_DETERMINISTIC=0 / PP_WGRAD_DETERMINISTIC=0 the
    atomic kernels run and only float-rounding agreement holds."""
    import bench
    from panopticsegforlargescalepointcloud_amd import ops
    from panopticsegforlargescalepointcloud_amd.applications import Data
    s = setup
    dev = torch.device("cuda")
    assert ops.WGRAD_DETERMINISTIC is True
    scene, b = s["scene"], s["b"]
    oid = b["origin_id"]
    inst = scene.inst[oid]
    inst_local = np.zeros_like(inst)
    for t in np.unique(b["batch"]):
        m = (b["batch"] == t) & (inst > 0)
        _, inv = np.unique(inst[m], return_inverse=True)
        inst_local[m] = inv + 1
    vote = (scene.inst_center[inst] - scene.pos[oid]).astype(np.float32)
    data = Data(pos=torch.from_numpy(b["pos"]), coords=torch.from_numpy(b["coords"]), x=torch.from_numpy(b["x"]),
                batch=torch.from_numpy(b["batch"]), y=torch.from_numpy(scene.cls[oid]),
                instance_labels=torch.from_numpy(inst_local), instance_mask=torch.from_numpy(inst > 0),
                vote_label=torch.from_numpy(vote), center_label=torch.from_numpy(scene.inst_center[inst]),
                num_instances=torch.tensor([int(inst_local.max())]))
    model = bench.build_model(dev, 0.05)[0].train()
    for m_ in model.modules():
        if isinstance(m_, torch.nn.BatchNorm1d):
            m_.momentum = 0.0                                  # (the two passes must see the same running statistics)
    runs = []
    for rep in range(2):
        model.set_input(data, dev)
        model.zero_grad(set_to_none=True)
        model.forward(epoch=1)
        model.backward(1)
        runs.append((model.loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    same = [n for n in runs[0][1] if torch.equal(runs[0][1][n], runs[1][1][n])]
    differ = [n for n in runs[0][1] if n not in same]
    print("loss %.7f vs %.7f (bits equal: %s); %d of %d gradient tensors bit-identical run to run" % (
        float(runs[0][0]), float(runs[1][0]), bool(torch.equal(runs[0][0], runs[1][0])), len(same), len(runs[0][1])))
    assert ops.SEGMENT_DETERMINISTIC is True
    assert torch.equal(runs[0][0], runs[1][0]), "loss bits differ run to run"
    assert not differ, "gradients differ run to run: %s" % differ[:8]


def test_bf16_conv_autocast_training_step_close_to_fp32():
    """ME.conv_autocast(): one training step of the full model with bfloat16 convolution compute gives a loss and
    gradients close to the fp32 step (bf16 operand rounding only), and differs from it (the bf16 kernels really ran)."""
    import bench
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    import train_microbench as tm
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    from panopticsegforlargescalepointcloud_amd.training import train_step
    dev = torch.device("cuda")
    scene, tiles, _ = bench.build_scene(40_000, 2, 0.05, 2022)
    data, n = tm.make_batch(scene, tiles, [0, 1])
    data = data.to(dev)
    out = {}
    for mode in ("fp32", "bf16"):
        model = bench.build_model(dev, 0.05)[0].train()
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        with ME.conv_autocast(mode == "bf16"):
            train_step(model, data, opt, 1, dev, 1)
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        out[mode] = (float(model.loss), grads)
    l32, g32 = out["fp32"]
    l16, g16 = out["bf16"]
    assert abs(l16 - l32) < 0.03 * abs(l32) and l16 != l32
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    assert 0.95 < cos < 1.0, cos  # measured 0.975: ~80 stacked convolutions with 2^-8 operand rounding each


def test_sparseconv3d_backend_block_matches_oracle():
    """a ResBlock-like stack written against the SparseConv3d.nn seam (snn.Conv3d / BatchNorm / ReLU / cat /
    Conv3dTranspose / SparseTensor, reference modules/SparseConv3d/modules.py:20-45,159) vs the oracle's convolutions."""
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd import sparseconv3d_nn as snn
    dev = torch.device("cuda")
    rng = np.random.default_rng(17)
    coords = bf.surface_coords(rng)                       # [N,4] (batch, x, y, z), unique
    n = len(coords)
    x = rng.normal(size=(n, 16)).astype(np.float32)
    torch.manual_seed(5)
    conv1, conv2 = snn.Conv3d(16, 32).to(dev), snn.Conv3d(32, 32, kernel_size=3, stride=2).to(dev)
    up = snn.Conv3dTranspose(32, 16, kernel_size=3, stride=2).to(dev)
    bn, relu = snn.BatchNorm(32).to(dev).eval(), snn.ReLU()
    with torch.no_grad():
        bn.bn.running_mean.normal_()
        bn.bn.running_var.uniform_(0.5, 2.0)
        st = snn.SparseTensor(torch.from_numpy(x), torch.from_numpy(coords[:, 1:]), torch.from_numpy(coords[:, 0]), device=dev)
        y1 = relu(bn(conv1(st)))
        y2 = conv2(y1)
        y3 = snn.cat(up(y2), st)
        got1, got2, got3 = y1.F.cpu().numpy(), y2.F.cpu().numpy(), y3.F.cpu().numpy()
        coarse_got = y2.C.cpu().numpy()
    scale = (bn.bn.weight / torch.sqrt(bn.bn.running_var + bn.bn.eps)).detach().cpu().numpy()
    shift = (bn.bn.bias - bn.bn.running_mean * torch.from_numpy(scale).to(dev)).detach().cpu().numpy()
    nbr = oracle.kernel_map(coords, coords, 3, 1, 1)
    want1 = oracle.spconv_fwd(x, conv1.kernel.detach().cpu().numpy(), nbr, n, scale=scale, shift=shift, relu=True)
    np.testing.assert_allclose(got1, want1, rtol=1e-4, atol=1e-4)
    coarse, _ = oracle.stride_coords(coords, 2)
    # the coarse level's row order is internal: compare as sets of (coordinate, feature row)
    down = oracle.kernel_map(coarse, coords, 3, 1, 1)
    want2 = oracle.spconv_fwd(want1, conv2.kernel.detach().cpu().numpy(), down, len(coarse))
    key = lambda c: np.lexsort(c.T[::-1])
    og, ow = key(coarse_got), key(coarse)
    assert np.array_equal(coarse_got[og], coarse[ow])
    np.testing.assert_allclose(got2[og], want2[ow], rtol=1e-4, atol=1e-4)
    upm = oracle.kernel_map(coords, coarse, 3, 1, -1)
    want3 = np.concatenate([oracle.spconv_fwd(want2, up.kernel.detach().cpu().numpy(), upm, n), x], 1)
    np.testing.assert_allclose(got3, want3, rtol=1e-4, atol=2e-4)


def test_side_stream_overlap_does_not_change_results(setup):
    """the worker-thread / side-stream stages (mean shift next to region growing, level + kernel-map prefetch) only
    reorder work: labels, proposals, scores and logits are bit-identical with both switched off, and stay identical over
    repeated passes (the prefetch plan is recorded on a model's first pass and replayed afterwards)."""
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    from panopticsegforlargescalepointcloud_amd.panoptic import pointgroup3heads as pg
    s = setup
    dev = torch.device("cuda")
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])

    def run():
        labels, res, counts = s["runner"].run(s["b"], len(s["ids"]), override=ov)
        torch.cuda.synchronize()
        return (labels.clone(), res.clusters_csr.offsets.clone(), res.clusters_csr.points.clone(),
                res.cluster_scores.clone(), res.semantic_logits.clone(), list(counts))

    saved = (ME.MAP_PREFETCH, pg.OVERLAP_CLUSTERING)
    try:
        ME.MAP_PREFETCH, pg.OVERLAP_CLUSTERING = True, True
        on = [run() for _ in range(3)]          # pass 1 may record the plan, passes 2-3 replay it
        ME.MAP_PREFETCH, pg.OVERLAP_CLUSTERING = False, False
        off = run()
    finally:
        ME.MAP_PREFETCH, pg.OVERLAP_CLUSTERING = saved
    for got in on:
        for a, b in zip(got[:5], off[:5]):
            assert torch.equal(a, b)
        assert got[5] == off[5]


def test_inference_switches_do_not_change_results(setup, monkeypatch):
    """The inference-path fusions and the early start of the map builder are pure scheduling: with each of them off the step
    returns bit-identical network outputs, proposals, scores and labels (fused heads vs one launch per head on
    caller-ordered features, fused 1x1 shortcuts vs their own launches, builder thread started in the coordinate
    manager's constructor vs at the first layer, mean shift next to region growing vs after it, compact vs dense same-level maps
    in the convolutions' prologue, scorer front end from the library vs tensor-library ops)."""
    from panopticsegforlargescalepointcloud_amd import applications, modules
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME
    from panopticsegforlargescalepointcloud_amd.panoptic import pointgroup3heads as pg
    s = setup
    dev = torch.device("cuda")
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])
    monkeypatch.setattr(ME, "COMPACT_MIN_ROWS", 1)  # (every same-level map of this small batch in its compact form too)

    def run():
        outs = []
        for _ in range(2):  # the second pass replays the map plan of the first (prefetch on the side stream)
            labels, res, counts = s["runner"].run(s["b"], len(s["ids"]), override=ov)
            outs.append((labels.clone(), res.semantic_logits.clone(), res.offset_logits.clone(), res.embed_logits.clone(),
                         res.cluster_scores.clone(), res.clusters_csr.offsets.clone(), res.clusters_csr.points.clone(), list(counts)))
        return outs

    ref = run()
    for mod, name in [(pg, "FUSE_HEADS"), (modules, "FUSE_SHORTCUT"), (applications, "EARLY_PREFETCH"), (pg, "OVERLAP_CLUSTERING"),
                      (ME, "COMPACT_MAPS"), (pg, "DEDUPE_FUSED")]:
        default = name != "COMPACT_MAPS"  # (the compact maps are off by default: measured slower)
        assert getattr(mod, name) is default
        monkeypatch.setattr(mod, name, not default)
        got = run()
        monkeypatch.setattr(mod, name, default)
        for r, g in zip(ref, got):
            for a, b in zip(r[:-1], g[:-1]):
                assert torch.equal(a, b), name
            assert r[-1] == g[-1], name


def test_proposal_deduplication_does_not_change_scores(setup):
    """Eval-time de-duplication of identical proposals (pointgroup3heads.py `dedupe_proposals`: region growing and mean shift
    often return the SAME point set; one representative per set goes through the ScorerUnet) is a pure saving: a proposal's
    score depends on its own rows only -- every output row of a convolution is summed in a fixed per-row order whatever else
    is in the batch, the per-proposal maximum is order-free -- so scores, labels and counts are BIT-identical with it off."""
    s = setup
    dev = torch.device("cuda")
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])
    model = s["model"]
    assert model.dedupe_proposals is True
    labels1, res1, counts1 = s["runner"].run(s["b"], len(s["ids"]), override=ov)
    sizes = res1.clusters_csr.sizes().cpu().numpy()
    pts, off = res1.clusters_csr.points.cpu().numpy(), res1.clusters_csr.offsets.cpu().numpy()
    sets = {}
    for i in range(len(sizes)):
        sets.setdefault(pts[off[i]:off[i + 1]].tobytes(), []).append(i)
    n_dup = sum(len(v) - 1 for v in sets.values())
    print("%d proposals, %d of them duplicates of an earlier one" % (len(sizes), n_dup))
    assert n_dup > 0, "no identical proposals in this batch: the test is void"
    model.dedupe_proposals = False
    try:
        labels0, res0, counts0 = s["runner"].run(s["b"], len(s["ids"]), override=ov)
    finally:
        model.dedupe_proposals = True
    assert torch.equal(res0.clusters_csr.points, res1.clusters_csr.points) and torch.equal(res0.clusters_csr.offsets, res1.clusters_csr.offsets)
    assert torch.equal(res0.cluster_scores, res1.cluster_scores)
    assert torch.equal(labels0, labels1) and list(counts0) == list(counts1)
    for group in sets.values():       # identical point sets carry identical scores (what the de-duplication relies on)
        sc = res0.cluster_scores[torch.tensor(group, device=dev)]
        assert bool((sc == sc[0]).all())


@pytest.mark.parametrize("scorer_type", ["MLP", "encoder"])
def test_other_scorer_types_match_oracle(setup, scorer_type):
    """scorer_type "MLP" (per-point ScorerMLP + per-proposal maximum) and "encoder" (ScorerEncoder: sparse down path +
    global max-pool head) of PointGroup3heads._compute_score (reference :419-426) against the oracle restatement; the MLP
    variant -- pure torch in the reference -- also against the same expression evaluated with plain torch modules."""
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    s = setup
    dev = torch.device("cuda")
    m = s["model"]
    ov = tuple(torch.from_numpy(a).to(dev) for a in s["override"])
    old = m._scorer_type
    g = torch.Generator().manual_seed(17)
    try:
        m._scorer_type = scorer_type
        with torch.no_grad():  # non-trivial BatchNorm statistics in the (otherwise never trained) scorer variants
            for name, buf in list(m.ScorerMLP.named_buffers()) + list(m.ScorerEncoder.named_buffers()):
                if name.endswith("running_mean"):
                    buf.copy_(torch.randn(buf.shape, generator=g).to(dev) * 0.1)
                elif name.endswith("running_var"):
                    buf.copy_((torch.rand(buf.shape, generator=g) + 0.5).to(dev))
        labels, res, counts = s["runner"].run(s["b"], len(s["ids"]), override=ov)
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        opt = {"cluster_radius_search": s["cfg"].cluster_radius_search, "cluster_type": s["cfg"].cluster_type,
               "bandwidth": s["cfg"].bandwidth}
        want = opipe.forward(sd, s["b"], opt, 9, syn.NPM3D_STUFF, override=s["override"], scorer_type=scorer_type)
        got = [c.cpu().numpy() for c in res.clusters_csr.to_list()]
        assert len(got) == len(want["clusters"]) and len(got) > 4
        assert _err("scores, scorer_type " + scorer_type, res.cluster_scores.cpu().numpy(), want["cluster_scores"]) < 1e-4
        assert float(res.cluster_scores.std()) > 0
        if scorer_type == "MLP":
            feats = torch.from_numpy(want["features"]).to(dev)
            rows = torch.cat([torch.from_numpy(c).to(dev) for c in want["clusters"]])
            b = torch.cat([torch.full((len(c),), i, device=dev) for i, c in enumerate(want["clusters"])])
            y = m.ScorerMLP(feats[rows])
            cf = torch.stack([y[b == i].max(0)[0] for i in range(len(want["clusters"]))])
            plain = m.ScorerHead(cf).squeeze(-1)
            assert _err("scores, MLP vs plain torch", res.cluster_scores.cpu().numpy(), plain.detach().cpu().numpy()) < 1e-4
    finally:
        m._scorer_type = old


@pytest.mark.parametrize("block", ["BottleneckBlock", "SEBlock", "SEBottleneckBlock"])
def test_other_block_types_match_a_dense_torch_composition(block):
    """the block types a backbone YAML can select besides ResBlock (api_modules.py:85-232), in inference (fused launches)
    and in training (autograd nodes, batch-statistics BN, differentiable SE pooling): outputs and input gradients against
    the same layers written with torch index arithmetic on the manager's kernel map"""
    import torch.nn.functional as F
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, modules as M
    rng = np.random.default_rng(12)
    coords = torch.from_numpy(bf.surface_coords(rng, n_batch=3, n=1500, extent=30)).cuda()
    torch.manual_seed(6)
    blk = getattr(M, block)(16, 32, ME.MinkowskiConvolution).cuda()
    for m in blk.modules():   # non-trivial running statistics and affine parameters
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)

    def dense(x, cm, training):
        nbr = cm.kernel_map_rows(1, 1, 3, 1).long()
        batch = cm.level(1).coords[:, 0].long()

        def conv(h, w):
            if w.dim() == 2:
                return h @ w
            out = torch.zeros(h.shape[0], w.shape[2], device="cuda")
            for k in range(27):
                ok = nbr[k] >= 0
                out = out.index_add(0, torch.nonzero(ok).view(-1), h[nbr[k][ok]] @ w[k])
            return out

        def chain(seq, h):
            for m in seq:
                if isinstance(m, ME.MinkowskiBatchNorm):
                    b = m.bn
                    h = F.batch_norm(h, b.running_mean.clone(), b.running_var.clone(), b.weight, b.bias, training, 0.1, b.eps)
                elif isinstance(m, ME.MinkowskiReLU):
                    h = torch.relu(h)
                else:
                    h = conv(h, m.kernel)
            return h
        out = chain(blk.block, x)
        if hasattr(blk, "SE"):
            nb = int(batch.max()) + 1
            mean = torch.zeros(nb, out.shape[1], device="cuda").index_add(0, batch, out) / torch.bincount(batch, minlength=nb)[:, None]
            fc = blk.SE.fc
            gate = torch.sigmoid(F.linear(torch.relu(F.linear(mean, fc[0].linear.weight, fc[0].linear.bias)),
                                          fc[2].linear.weight, fc[2].linear.bias))
            out = out * gate[batch]
        return out + (chain(blk.downsample, x) if blk.downsample else x)

    x0 = torch.randn(len(coords), 16, device="cuda")
    for training in (False, True):
        blk.train(training)
        x = x0.clone().requires_grad_(training)
        with torch.set_grad_enabled(training):
            st = ME.SparseTensor(features=x, coordinates=coords, device="cuda")
            y = blk(st)
        cm = st.coordinate_manager
        x2 = x0.clone().requires_grad_(training)
        xin = x2 if cm.perm is None else x2[cm.perm]
        with torch.set_grad_enabled(training):
            want = dense(xin, cm, training)
        got = y.feats   # internal row order, like `want`
        bf.scaled_err("%s %s" % (block, "train" if training else "eval"), got.detach().cpu().numpy(), want.detach().cpu().numpy())
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
        if training:
            (y.F ** 2).mean().backward()
            (want ** 2).mean().backward()
            np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-3, atol=1e-6)
