"""Host profile of the STEADY-STATE step only (bench.py's set-up, priming and warm-up are outside the profile): where the
Python side of one TileRunner pass spends its time on a dispatch-paced configuration (one rank's share at 8 GPUs).
usage (GPU box): python profiles/host_step_profile.py [points=1250000] [grid=3] [steps=20]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.scene import TileRunner  # noqa: E402

points = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
scene, tiles, _ = bench.build_scene(points, grid, 0.05, 2022)
model, cfg, DS = bench.build_model(dev, 0.05)
runner = TileRunner(model, dev, backbone_ahead=os.environ.get("PP_BACKBONE_AHEAD", "1") != "0")
ids = list(range(len(tiles)))
b = syn.tile_batch(scene, tiles, ids)
ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(2022))
dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
ovd = tuple(torch.from_numpy(a).to(dev) for a in ov)
for _ in range(5):
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b, after_next=dev_b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b, after_next=dev_b)
torch.cuda.synchronize()
print("unprofiled: %.2f ms per step (%d voxels, %d tiles)" % (1e3 * (time.perf_counter() - t0) / steps, len(b["pos"]), len(ids)))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    runner.run(dev_b, len(ids), override=ovd, next_batch=dev_b, after_next=dev_b)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print("(per step = / %d; the main thread only: the two builder threads are not profiled)" % steps)
print(s.getvalue()[:8000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(35)
print(s.getvalue()[:7000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_callers(r"method '(to|tolist|item|cpu)' of")
print(s.getvalue()[:6000])
