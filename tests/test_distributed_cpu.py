"""world_size-2 gloo test of the multi-GPU path on CPU tensors: tile sharding, the single all-gather exchange of
per-tile label arrays, and the order-dependent scene assembly being identical on every rank and equal to the
single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tile_results(n_tiles, n_scene, with_votes=False):
    """deterministic fake per-tile outputs: (origin_ids, labels)"""
    out = {}
    for t in range(n_tiles):
        rng = np.random.default_rng(100 + t)
        start = t * 700
        origin = np.sort(rng.choice(np.arange(start, min(start + 1500, n_scene)), size=900 + 37 * t, replace=False))
        labels = np.full(len(origin), -1, np.int32)
        k = 4 + t % 3
        cuts = np.sort(rng.choice(len(origin), size=k + 1, replace=False))
        for i in range(k):
            labels[cuts[i]: cuts[i + 1]] = i
        out[t] = (torch.from_numpy(origin.astype(np.int64)), torch.from_numpy(labels))
        if with_votes:
            out[t] = out[t] + (torch.from_numpy(rng.normal(size=(len(origin), 9)).astype(np.float32)),)
    return out


def _assemble(results, n_scene):
    from panopticsegforlargescalepointcloud_amd.scene import assemble_scene
    asm = assemble_scene(results, sorted(results), n_scene, 9)  # original block order (the greedy merge is order dependent)
    return asm.ins_pre, asm.max_instance, asm.prediction_count, asm.votes


def _worker(rank, world, port, n_tiles, n_scene, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.scene import exchange_tile_results, shard_tiles
    allr = _tile_results(n_tiles, n_scene, with_votes=True)
    sizes = [len(allr[t][0]) for t in range(n_tiles)]
    mine = shard_tiles(sizes, world)[rank]
    local = {t: allr[t] for t in mine}
    full = exchange_tile_results(local)
    ok = sorted(full) == list(range(n_tiles)) and all(
        torch.equal(full[t][0], allr[t][0]) and torch.equal(full[t][1], allr[t][1]) and torch.equal(full[t][2], allr[t][2])
        for t in range(n_tiles))
    ins, mx, cnt, votes = _assemble(full, n_scene)
    q.put((rank, ok, ins.tobytes(), mx, int(cnt.sum()), votes.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_and_assembly_world2():
    n_tiles, n_scene, world = 9, 8000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tiles, n_scene, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: votes[origin] += log-probs, prediction_count += 1, block merging in block order
    ref_ins, ref_mx, ref_cnt, ref_votes = _assemble(_tile_results(n_tiles, n_scene, with_votes=True), n_scene)
    assert np.abs(ref_votes).sum() > 0
    for rank, ok, ins_bytes, mx, cnt, votes_bytes in got:
        assert ok, "rank %d did not receive every tile intact" % rank
        assert ins_bytes == ref_ins.tobytes() and mx == ref_mx and cnt == int(ref_cnt.sum())
        assert votes_bytes == ref_votes.tobytes(), "semantic votes of the assembled scene differ on rank %d" % rank


def _worker_edge(rank, world, port, q):
    """one rank without tiles, origin ids beyond 2^31 (the exchange switches to an int64 buffer)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.scene import exchange_tile_results
    big = 2 ** 31 + 5
    g = torch.Generator().manual_seed(1)
    allr = {3: (torch.arange(big, big + 40, dtype=torch.int64), torch.arange(40, dtype=torch.int32) % 7 - 1,
                torch.randn(40, 9, generator=g))}
    local = dict(allr) if rank == 0 else {}
    full = exchange_tile_results(local)
    ok = sorted(full) == [3] and torch.equal(full[3][0], allr[3][0]) and torch.equal(full[3][1], allr[3][1])
    ok = ok and full[3][0].dtype == torch.int64 and full[3][1].dtype == torch.int32 and torch.equal(full[3][2], allr[3][2])
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_empty_rank_and_wide_ids_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in got), got


def test_exchange_single_process_is_identity():
    from panopticsegforlargescalepointcloud_amd.scene import exchange_tile_results
    r = _tile_results(3, 4000)
    out = exchange_tile_results(r)
    assert sorted(out) == [0, 1, 2] and all(torch.equal(out[t][1], r[t][1]) for t in r)


# ---------------------------------------------------------------- data-parallel gradient averaging (config C5)
def _dp_model():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(),
                               torch.nn.Linear(300, 7), torch.nn.Linear(7, 3))  # last layer: never used -> no gradient


def _dp_batch(rank):
    g = torch.Generator().manual_seed(50 + rank)
    return torch.randn(64, 40, generator=g), torch.randn(64, 7, generator=g)


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.training import allreduce_gradients
    model = _dp_model()
    x, y = _dp_batch(rank)
    loss = torch.nn.functional.mse_loss(model[:5](x), y)
    loss.backward()
    if rank == 1:  # a parameter that got a gradient on ONE rank only must still be reduced on both
        model[5].weight.grad = torch.ones_like(model[5].weight)
    n_buckets = allreduce_gradients(list(model.parameters()), world, bucket_bytes=200_000)
    assert model[5].bias.grad is None  # no rank produced a gradient for it: stays None (the optimizer skips it)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    q.put((rank, n_buckets, flat.numpy().tobytes()))
    # overlapped form: hooks launch the buckets during backward; parameters without a gradient on ANY rank keep grad None
    from panopticsegforlargescalepointcloud_amd.training import GradientReducer
    model2 = _dp_model()
    reducer = GradientReducer(model2.parameters(), bucket_bytes=200_000)
    torch.nn.functional.mse_loss(model2[:5](x), y).backward()
    nb2 = reducer.finish()  # first step: the never-used layer sits in the first bucket (reverse order) and blocks it
    unused_none = model2[5].weight.grad is None and model2[5].bias.grad is None
    flat2 = torch.cat([p.grad.reshape(-1) for p in list(model2.parameters())[:-2]])
    # second step through the same reducer (state is reset by finish)
    for p in model2.parameters():
        p.grad = None
    before = reducer.launched_in_backward
    torch.nn.functional.mse_loss(model2[:5](x), y).backward()
    in_backward = reducer.launched_in_backward - before  # ... from the second step on it trails and the others overlap
    reducer.finish()
    flat3 = torch.cat([p.grad.reshape(-1) for p in list(model2.parameters())[:-2]])
    q.put((rank, "hooks", nb2, in_backward, unused_none, flat2.numpy().tobytes(), bool(torch.equal(flat2, flat3))))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2():
    from panopticsegforlargescalepointcloud_amd.training import gradient_buckets
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=120) for _ in range(2 * world)]
    got = sorted(m for m in msgs if m[1] != "hooks")
    hooks = sorted((m for m in msgs if m[1] == "hooks"), key=lambda m: m[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: mean of the two ranks' gradients (missing gradients count as zeros)
    grads = []
    for rank in range(world):
        model = _dp_model()
        x, y = _dp_batch(rank)
        torch.nn.functional.mse_loss(model[:5](x), y).backward()
        if rank == 1:
            model[5].weight.grad = torch.ones_like(model[5].weight)
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]))
    want = ((grads[0] + grads[1]) / 2).numpy()
    for rank, n_buckets, blob in got:
        np.testing.assert_allclose(np.frombuffer(blob, np.float32), want, rtol=1e-6, atol=1e-7)
        assert n_buckets == len(gradient_buckets(list(_dp_model().parameters())[::-1], 200_000)) and n_buckets >= 2
    n_used = sum(p.numel() for p in list(_dp_model().parameters())[:-2])
    for rank, _, nb2, in_backward, unused_none, blob, same_again in hooks:
        assert nb2 >= 2 and unused_none and same_again
        assert in_backward >= 1, "no bucket was launched before backward returned (no overlap)"
        # used parameters: mean of the two ranks' gradients (model[5] got no gradient anywhere in this run)
        grads2 = []
        for r in range(world):
            m = _dp_model()
            xx, yy = _dp_batch(r)
            torch.nn.functional.mse_loss(m[:5](xx), yy).backward()
            grads2.append(torch.cat([p.grad.reshape(-1) for p in list(m.parameters())[:-2]]))
        np.testing.assert_allclose(np.frombuffer(blob, np.float32), ((grads2[0] + grads2[1]) / 2).numpy(), rtol=1e-6, atol=1e-7)
        assert len(np.frombuffer(blob, np.float32)) == n_used


# ---------------------------------------------------------------- hook-based reducer with asymmetric gradients
def _asym_worker(rank, world, port, q):
    """rank 1 does not use the middle layer (the scorer of a rank whose batch produced no proposals): it completes fewer
    buckets during backward than rank 0, so the collectives finish() adds must sit at a rank-independent position."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.training import GradientReducer
    torch.manual_seed(9)
    a, mid, c = torch.nn.Linear(30, 200), torch.nn.Linear(200, 200), torch.nn.Linear(200, 5)
    params = list(a.parameters()) + list(mid.parameters()) + list(c.parameters())
    reducer = GradientReducer(params, bucket_bytes=100_000)  # >= 3 buckets: c | mid | a (reverse registration order)
    g = torch.Generator().manual_seed(70 + rank)
    x, y = torch.randn(32, 30, generator=g), torch.randn(32, 5, generator=g)
    launched = []
    for step in range(2):  # second step: the bucket layout has been re-sorted by the reduced mask
        for p in params:
            p.grad = None
        before = reducer.launched_in_backward
        h = torch.relu(a(x))
        if rank == 0:
            h = torch.relu(mid(h))
        torch.nn.functional.mse_loss(c(h), y).backward()
        launched.append(reducer.launched_in_backward - before)
        reducer.finish()
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    q.put((rank, launched, flat.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_hooks_asymmetric_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_asym_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks end with the same averaged gradients; rank 0 launched more buckets inside backward than rank 1 did
    assert msgs[0][2] == msgs[1][2]
    assert msgs[0][1][0] > msgs[1][1][0], msgs
    grads = []
    for rank in range(world):
        torch.manual_seed(9)
        a, mid, c = torch.nn.Linear(30, 200), torch.nn.Linear(200, 200), torch.nn.Linear(200, 5)
        g = torch.Generator().manual_seed(70 + rank)
        x, y = torch.randn(32, 30, generator=g), torch.randn(32, 5, generator=g)
        h = torch.relu(a(x))
        if rank == 0:
            h = torch.relu(mid(h))
        torch.nn.functional.mse_loss(c(h), y).backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in list(a.parameters()) + list(mid.parameters()) + list(c.parameters())]))
    np.testing.assert_allclose(np.frombuffer(msgs[0][2], np.float32), ((grads[0] + grads[1]) / 2).numpy(), rtol=1e-6, atol=1e-7)
