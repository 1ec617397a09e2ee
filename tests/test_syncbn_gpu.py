"""GPU: SyncBN for data-parallel training.  The reference normalises over all 4 cylinders of its batch on ONE GPU
(conf/training/7_area1.yaml:5); with the batch sharded over ranks, training.enable_sync_bn all-reduces the per-channel sums so
that 2 ranks x 2 cylinders use the statistics of the whole batch: same normalised outputs, same running statistics, same input
gradients as 1 rank x 4 cylinders.  Two ranks share the one GPU of the test box (gloo collectives)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = [0, 1, 2, 3]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _op_case():
    rng = np.random.default_rng(5)
    n, c = 30011, 48
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3.0, c) + rng.normal(size=c)).astype(np.float32)
    dy = rng.normal(size=(n, c)).astype(np.float32)
    w, b = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    return x, dy, w, b, n // 3  # (an uneven split: rank 0 holds a third of the rows)


def _model_and_batch(ids):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import bench
    import train_microbench as tm
    dev = torch.device("cuda", 0)
    scene, tiles, _ = bench.build_scene(60_000, 2, 0.05, 2022)
    data, n = tm.make_batch(scene, tiles, ids)
    model = bench.build_model(dev, 0.05)[0].train()
    return model, data.to(dev), dev


def _forward_stats(model, data, dev):
    model.set_input(data, dev)
    model.forward(epoch=1)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    return (model.output.semantic_logits.detach().cpu().numpy(), bns[0].running_mean.cpu().numpy().copy(),
            bns[5].running_var.cpu().numpy().copy(), float(model.semantic_loss.detach()) if hasattr(model, "semantic_loss") else None)


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from panopticsegforlargescalepointcloud_amd import ops, training
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        training.enable_sync_bn(True)
        # ---- the operator on an uneven split of one tensor
        x, dy, w, b, cut = _op_case()
        lo, hi = (0, cut) if rank == 0 else (cut, len(x))
        rm, rv = torch.zeros(x.shape[1], device=dev), torch.ones(x.shape[1], device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        xs, dys = torch.from_numpy(x[lo:hi]).to(dev), torch.from_numpy(dy[lo:hi]).to(dev)
        wt, bt = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
        y, mean, rstd = ops.bn_train_fwd(xs, wt, bt, 1e-5, 0.1, rm, rv, True, nbt)
        dx, dw, db = ops.bn_train_bwd(xs, dys, y, wt, mean, rstd)
        op = [t.cpu().numpy() for t in (y, dx, dw, db, rm, rv)] + [int(nbt)]
        # ---- the model: this rank's two cylinders of the batch of four
        model, data, dev = _model_and_batch(TILES[2 * rank: 2 * rank + 2])
        fw = _forward_stats(model, data, dev)
        q.put((rank, op, fw[:3], ops.SYNC_BN_STATS["all_reduces"]))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "error: %r" % (e,)))
        raise


def test_two_ranks_normalise_like_one_rank_with_the_whole_batch():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sys.path.insert(0, ROOT)
    from panopticsegforlargescalepointcloud_amd import ops
    assert ops.SYNC_BN_GROUP is None
    dev = torch.device("cuda", 0)
    # ---- one rank, whole batch: the fused per-replica kernels
    x, dy, w, b, cut = _op_case()
    rm, rv = torch.zeros(x.shape[1], device=dev), torch.ones(x.shape[1], device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    xt, dyt, wt, bt = (torch.from_numpy(a).to(dev) for a in (x, dy, w, b))
    y, mean, rstd = ops.bn_train_fwd(xt, wt, bt, 1e-5, 0.1, rm, rv, True, nbt)
    dx, dw, db = ops.bn_train_bwd(xt, dyt, y, wt, mean, rstd)
    model, data, dev = _model_and_batch(TILES)
    sem, rm0, rv5, _ = _forward_stats(model, data, dev)
    n_first = int((data.batch < 2).sum())
    # ---- two ranks, half the batch each
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda m: m[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(len(m) == 4 for m in got), got
    (_, op0, fw0, ar0), (_, op1, fw1, ar1) = got
    # operator: outputs and input gradients of the rows each rank holds, summed weight / bias gradients, running statistics
    tol = dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np.concatenate([op0[0], op1[0]]), y.cpu().numpy(), **tol)
    np.testing.assert_allclose(np.concatenate([op0[1], op1[1]]), dx.cpu().numpy(), **tol)
    np.testing.assert_allclose(op0[2] + op1[2], dw.cpu().numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(op0[3] + op1[3], db.cpu().numpy(), rtol=1e-4, atol=1e-3)
    for o in (op0, op1):
        np.testing.assert_allclose(o[4], rm.cpu().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(o[5], rv.cpu().numpy(), rtol=1e-6, atol=1e-7)
        assert o[6] == 1
    # model: per-point semantic log-probabilities of the two halves == the rows of the batch of four (hence any loss computed
    # from them), and the running statistics of the first and of a deeper BatchNorm
    both = np.concatenate([fw0[0], fw1[0]])
    assert len(fw0[0]) == n_first and both.shape == sem.shape
    err = float(np.abs(both - sem).max())
    print("max |semantic log-prob difference| 2 x 2 cylinders vs 1 x 4: %.2e" % err)
    assert err < 2e-5, err
    y_lab = data.y.cpu().numpy()
    valid = y_lab >= 0
    nll = lambda lp: float(-lp[valid, y_lab[valid]].mean())  # noqa: E731
    assert abs(nll(both) - nll(sem)) < 1e-6
    for fw in (fw0, fw1):
        np.testing.assert_allclose(fw[1], rm0, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(fw[2], rv5, rtol=1e-5, atol=1e-7)
    assert ar0 == ar1 and ar0 >= 80  # one all-reduce per training-mode BatchNorm and direction


def _train_worker(rank, world, port, q):
    """two steps of training.train_step past prepare_epoch with SyncBN and a GradientReducer; rank 1 finds no proposal at all
    (every class declared "stuff"), so it never runs the scorer -- the ranks' launch sequences differ, their collectives must not"""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from panopticsegforlargescalepointcloud_amd import ops, training
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        training.enable_sync_bn(True)
        model, data, dev = _model_and_batch(TILES[2 * rank: 2 * rank + 2])
        if rank == 1:
            model._stuff_classes = torch.arange(-1, model.num_classes)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
        reducer = training.GradientReducer(model.parameters(), bucket_bytes=8 << 20)
        epoch = int(model.opt.prepare_epoch) + 1
        losses, props = [], []
        for _ in range(2):
            losses.append(training.train_step(model, data, opt, epoch, dev, world, reducer=reducer))
            csr = model.output.clusters_csr
            props.append(0 if csr is None else int(csr.n))
        named = dict(model.named_parameters())
        scorer_grad = any(p.grad is not None for n_, p in named.items() if n_.startswith("ScorerUnet."))
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        q.put((rank, losses, props, scorer_grad, reducer.launched_in_backward, ops.SYNC_BN_STATS["all_reduces"], flat.tobytes()))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "error: %r" % (e,)))
        raise


def test_syncbn_training_with_a_rank_that_has_no_proposals():
    """the collective sequence must be the same on every rank.  Rank 0 groups and scores its proposals (ScorerUnet
    BatchNorms, score loss, scorer gradients), rank 1 has none: the scorer's BatchNorms stay per replica, the gradient buckets wait
    for finish(), both ranks issue the same number of BatchNorm all-reduces and end with identical parameters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = sorted((q.get(timeout=400) for _ in range(world)), key=lambda m: m[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:  # (a mismatched collective blocks its ranks: do not leave them behind)
            if p.is_alive():
                p.terminate()
    assert all(len(m) == 7 for m in got), got
    (_, l0, p0, sg0, in_bwd0, ar0, w0), (_, l1, p1, sg1, in_bwd1, ar1, w1) = got
    assert all(np.isfinite(l0 + l1))
    assert min(p0) > 0 and max(p1) == 0, (p0, p1)         # rank 0 scored proposals, rank 1 had none
    assert sg0 and sg1                                      # ... yet both hold the (averaged) scorer gradients
    assert in_bwd0 == 0 and in_bwd1 == 0                    # with SyncBN every bucket is launched by finish()
    assert ar0 == ar1 and ar0 >= 2 * 2 * 80                 # same BatchNorm all-reduces on both ranks (2 steps x 2 directions)
    assert w0 == w1, "replicas diverged"
