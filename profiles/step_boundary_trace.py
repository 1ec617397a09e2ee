"""Host-side time stamps around the step boundary of the bench configuration (no profiler attached): when does the scorer's
checked segment maximum return, when are NMS and the final host read issued / back, when is the next step's first convolution
launched.  usage (GPU box): python profiles/step_boundary_trace.py [steps=6]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import ops, scene as scene_mod, synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.scene import TileRunner  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda")
scene, tiles, _ = bench.build_scene(10_000_000, 8, 0.05, 2022)
model, cfg, DS = bench.build_model(dev, 0.05)
runner = TileRunner(model, dev)
ids = list(range(len(tiles)))
b = syn.tile_batch(scene, tiles, ids)
ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(2022))
dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
ovd = tuple(torch.from_numpy(a).to(dev) for a in ov)
log = []


def wrap(mod, name, tag):
    fn = getattr(mod, name)

    def inner(*a, **k):
        log.append((tag + " in", time.perf_counter()))
        out = fn(*a, **k)
        log.append((tag + " out", time.perf_counter()))
        return out
    setattr(mod, name, inner)


wrap(ops, "segment_reduce", "segment_reduce")
wrap(ops, "nms_paint", "nms_paint")
wrap(scene_mod, "instance_labels_per_tile", "labels_per_tile")
wrap(ops, "spconv_fwd", "conv")
wrap(ops, "region_grow_csr", "region_grow")
wrap(ops, "proposals_unique", "proposals_unique")
if os.environ.get("PP_TRACE_FRONT", "0") == "1":  # round 5: the scorer's front end, call by call (main thread AND builder threads)
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME_
    for nm in ("morton_order", "block_index_build", "block_index_coarsen", "kernel_map_bi", "map_order", "map_permute", "level_permute",
               "gather_rows", "kernel_map_transpose8", "proposal_pairs", "meanshift", "compose_perm"):
        wrap(ops, nm, nm)
    wrap(ME_, "_order_level", "_order_level")
for _ in range(3):
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b)
torch.cuda.synchronize()
del log[:]
marks = []
for _ in range(steps):
    marks.append(("step start", time.perf_counter()))
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b)
    marks.append(("step end", time.perf_counter()))
torch.cuda.synchronize()
ev = sorted(log + marks, key=lambda x: x[1])
# one step in the middle: everything relative to its start, convolutions collapsed to first / last of each run
t_start = [t for n, t in ev if n == "step start"][steps // 2]
t_next = [t for n, t in ev if n == "step start"][steps // 2 + 1] if steps // 2 + 1 < steps else 1e30
prev = None
n_conv = 0
for n, t in ev:
    if t < t_start or t > t_next + 0.02:
        continue
    if os.environ.get("PP_TRACE_FRONT", "0") == "1" and not (float(os.environ.get("PP_TRACE_LO", "70")) <= 1e3 * (t - t_start) <= float(os.environ.get("PP_TRACE_HI", "95"))):
        continue
    if n.startswith("conv"):
        n_conv += 1
        prev = (n, t)
        continue
    if n_conv:
        print("%9.3f ms   ... %d convolution launch events, last at %.3f" % (1e3 * (prev[1] - t_start), n_conv, 1e3 * (prev[1] - t_start)))
        n_conv = 0
    print("%9.3f ms   %s" % (1e3 * (t - t_start), n))
