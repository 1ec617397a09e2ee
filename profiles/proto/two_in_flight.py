"""Prototype (measurement only): the scene's 64 tiles as TWO half-batches in flight -- two model replicas, two host threads,
two streams -- against one 64-tile batch.  PP_CONV_SCRATCH_MB=0 (the split-K scratch is one global buffer)."""
import copy
import os
import sys
import threading
import time

os.environ.setdefault("PP_CONV_SCRATCH_MB", "0")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.scene import TileRunner  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME  # noqa: E402

dev = torch.device("cuda")
scene, tiles, _ = bench.build_scene(10_000_000, 8, 0.05, 2022)
model, cfg, DS = bench.build_model(dev, 0.05)
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
order = sorted(range(len(tiles)), key=lambda t: -len(tiles[t]))
parts = [sorted(order[i::NW]) for i in range(NW)]
rng = np.random.default_rng(2022)


def mk(ids):
    b = syn.tile_batch(scene, tiles, ids)
    ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, rng)
    return ({k: torch.from_numpy(v).to(dev) for k, v in b.items()}, tuple(torch.from_numpy(a).to(dev) for a in ov), len(ids))


full = mk(sorted(order))
halves = [mk(p) for p in parts]
if len(sys.argv) > 4 and sys.argv[4] == "full":   # NW whole scenes in flight (throughput of a stream of scenes)
    halves = [full for _ in range(NW)]
runner = TileRunner(model, dev)
for _ in range(3):
    runner.run(full[0], full[2], override=full[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    runner.run(full[0], full[2], override=full[1], next_batch=full[0])
torch.cuda.synchronize()
print("one batch of %d tiles: %.2f ms per scene" % (full[2], 1e3 * (time.perf_counter() - t0) / steps))
# sequential half batches
for _ in range(2):
    for h in halves:
        runner.run(h[0], h[2], override=h[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for j, h in enumerate(halves):
        runner.run(h[0], h[2], override=h[1], next_batch=halves[(j + 1) % NW][0])
torch.cuda.synchronize()
print("%d part batches, one after the other: %.2f ms per scene" % (NW, 1e3 * (time.perf_counter() - t0) / steps))

replicas = [model]
for _ in range(NW - 1):
    m2 = bench.build_model(dev, 0.05)[0]
    m2.load_state_dict(model.state_dict())
    replicas.append(m2)
runners = [TileRunner(m, dev) for m in replicas]
streams = [torch.cuda.Stream(device=dev) for _ in range(NW)]
sides = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(NW)]
out = [None] * NW
stagger = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0


def worker(i, n):
    with torch.cuda.device(dev), torch.cuda.stream(streams[i]):
        ME._SIDE_STREAMS_TLS = None
        if i and stagger:
            time.sleep(stagger * 1e-3)
        for _ in range(n):
            out[i] = runners[i].run(halves[i][0], halves[i][2], override=halves[i][1], next_batch=halves[i][0])
        streams[i].synchronize()


def go(n):
    th = [threading.Thread(target=worker, args=(i, n)) for i in range(NW)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()


go(3)
t0 = time.perf_counter()
go(steps)
per = NW if (len(sys.argv) > 4 and sys.argv[4] == "full") else 1
print("%d batches in flight (%d threads / streams): %.2f ms per scene" % (NW, NW, 1e3 * (time.perf_counter() - t0) / steps / per))
