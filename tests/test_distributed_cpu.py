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


def _tile_results(n_tiles, n_scene):
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
    return out


def _assemble(results, n_scene):
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler
    asm = SceneAssembler(n_scene, 9)
    for t in sorted(results):  # original block order (the greedy merge is order dependent)
        asm.add_block(results[t][0].numpy(), results[t][1].numpy())
    return asm.ins_pre, asm.max_instance, asm.prediction_count


def _worker(rank, world, port, n_tiles, n_scene, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.scene import exchange_tile_results, shard_tiles
    allr = _tile_results(n_tiles, n_scene)
    sizes = [len(allr[t][0]) for t in range(n_tiles)]
    mine = shard_tiles(sizes, world)[rank]
    local = {t: allr[t] for t in mine}
    full = exchange_tile_results(local)
    ok = sorted(full) == list(range(n_tiles)) and all(
        torch.equal(full[t][0], allr[t][0]) and torch.equal(full[t][1], allr[t][1]) for t in range(n_tiles))
    ins, mx, cnt = _assemble(full, n_scene)
    q.put((rank, ok, ins.tobytes(), mx, int(cnt.sum())))
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
    ref_ins, ref_mx, ref_cnt = _assemble(_tile_results(n_tiles, n_scene), n_scene)
    for rank, ok, ins_bytes, mx, cnt in got:
        assert ok, "rank %d did not receive every tile intact" % rank
        assert ins_bytes == ref_ins.tobytes() and mx == ref_mx and cnt == int(ref_cnt.sum())


def _worker_edge(rank, world, port, q):
    """one rank without tiles, origin ids beyond 2^31 (the exchange switches to an int64 buffer)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticsegforlargescalepointcloud_amd.scene import exchange_tile_results
    big = 2 ** 31 + 5
    allr = {3: (torch.arange(big, big + 40, dtype=torch.int64), torch.arange(40, dtype=torch.int32) % 7 - 1)}
    local = dict(allr) if rank == 0 else {}
    full = exchange_tile_results(local)
    ok = sorted(full) == [3] and torch.equal(full[3][0], allr[3][0]) and torch.equal(full[3][1], allr[3][1])
    ok = ok and full[3][0].dtype == torch.int64 and full[3][1].dtype == torch.int32
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
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    q.put((rank, n_buckets, flat.numpy().tobytes()))
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
    got = sorted(q.get(timeout=120) for _ in range(world))
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
        assert n_buckets == len(gradient_buckets(list(_dp_model().parameters()), 200_000)) and n_buckets >= 2
