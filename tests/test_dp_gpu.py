"""GPU: data-parallel training of the REAL model with two ranks sharing one GPU (gloo collectives, the backend the
single-GPU box offers): replicas stay bit-identical, the bucketed all-reduce is launched during backward, parameters
without a gradient on any rank keep grad None."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bench
        import train_microbench as tm
        from panopticsegforlargescalepointcloud_amd.training import GradientReducer, train_step
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        scene, tiles, _ = bench.build_scene(40_000, 2, 0.05, 2022)
        data, n = tm.make_batch(scene, tiles, [rank, (rank + 2) % len(tiles)])  # every rank trains on its own cylinders
        data = data.to(dev)
        model = bench.build_model(dev, 0.05)[0].train()  # same seed on every rank => identical replicas
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
        reducer = GradientReducer(model.parameters(), bucket_bytes=8 << 20)
        losses, overlap = [], []
        for it in range(3):
            before = reducer.launched_in_backward
            losses.append(train_step(model, data, opt, 1, dev, world, reducer=reducer))  # epoch 1 <= prepare_epoch: the scorer is unused
            overlap.append(reducer.launched_in_backward - before)
        unused = [n_ for n_, p in model.named_parameters() if n_.startswith(("ScorerUnet.", "ScorerEncoder.", "ScorerMLP."))]
        none_grads = all(dict(model.named_parameters())[n_].grad is None for n_ in unused)
        no_state = all(len(opt.state.get(dict(model.named_parameters())[n_], {})) == 0 for n_ in unused)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        q.put((rank, losses, overlap, len(reducer.buckets), none_grads and no_state and len(unused) > 50, flat.tobytes()))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:  # surface the failure instead of a queue timeout
        q.put((rank, "error: %r" % (e,)))
        raise


def test_two_replicas_on_one_gpu_stay_identical():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
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
    assert all(len(m) == 6 for m in got), got
    (r0, l0, ov0, nb0, ok0, w0), (r1, l1, ov1, nb1, ok1, w1) = got
    assert w0 == w1, "replicas diverged"
    assert ok0 and ok1, "never-used parameters must keep grad None and get no optimizer state"
    assert l0 != l1 and all(np.isfinite(l0 + l1))           # different data per rank, finite losses
    assert nb0 == nb1 and nb0 >= 3
    # first step: the never-used scorer parameters sit in the leading buckets (reverse registration order) and hold them
    # back; from the second step on they trail and the buckets of the live parameters are reduced DURING backward
    assert ov0[1] >= 2 and ov1[1] >= 2 and ov0[2] >= 2, (ov0, ov1)
