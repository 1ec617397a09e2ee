"""Which lines of an inference pass (one rank's share at 8 GPUs by default) synchronise with the GPU, as torch sees them: one
TileRunner pass under torch.cuda.set_sync_debug_mode("warn"), warnings grouped by the first frame inside this package and by thread.
(Reads issued from C -- region growing, mean shift -- are not seen by torch.)
usage (GPU box): python profiles/infer_sync_trace.py [points=1250000] [grid=3]"""
import collections
import os
import sys
import threading
import traceback
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.scene import TileRunner  # noqa: E402

points = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
scene, tiles, _ = bench.build_scene(points, grid, 0.05, 2022)
model, cfg, DS = bench.build_model(dev, 0.05)
runner = TileRunner(model, dev)
ids = list(range(len(tiles)))
b = syn.tile_batch(scene, tiles, ids)
ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(2022))
dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
ovd = tuple(torch.from_numpy(a).to(dev) for a in ov)
for _ in range(4):
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b)
torch.cuda.synchronize()
sites = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    th = threading.current_thread().name
    for fr in reversed(traceback.extract_stack()):
        if "panopticsegforlargescalepointcloud_amd" in fr.filename:
            sites[(th, os.path.basename(fr.filename), fr.lineno, (fr.line or "").strip()[:80])] += 1
            return
    sites[(th, "?", 0, str(message)[:60])] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("default")
print("%d synchronising torch calls in one pass (%d voxels, %d tiles)" % (sum(sites.values()), len(b["pos"]), len(ids)))
for (th, f, l, src), c in sorted(sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%3d  %-18s %s:%d  %s" % (c, th, f, l, src))
