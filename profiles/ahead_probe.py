"""probe: backbone_ahead switched on after serial steps vs from the start (per-step wall times on the bench scene)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from panopticsegforlargescalepointcloud_amd import synthetic as syn
from panopticsegforlargescalepointcloud_amd.scene import TileRunner

dev = torch.device("cuda")
scene, tiles, radius = bench.build_scene(10_000_000, 8, 0.05, 2022)
model, cfg, DS = bench.build_model(dev, 0.05)
ids = list(range(len(tiles)))
b = syn.tile_batch(scene, tiles, ids)
cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(2022))
dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
ov = (torch.from_numpy(cls).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(emb).to(dev))
runner = TileRunner(model, dev, backbone_ahead=(sys.argv[1] == "start"))


def steps(n, tag):
    out = []
    for _ in range(n):
        t = time.perf_counter()
        runner.run(dev_b, len(ids), override=ov, next_batch=dev_b, after_next=dev_b)
        out.append(round(1e3 * (time.perf_counter() - t), 1))
    torch.cuda.synchronize()
    print(tag, out)


steps(6, "phase 1 (ahead=%s)" % runner.backbone_ahead)
if sys.argv[1] == "switch":
    for _ in range(3):
        runner.run(dev_b, len(ids), override=ov)   # (no next batch: as bench's single-scene steps)
    runner.backbone_ahead = True
steps(8, "phase 2 (ahead=%s)" % runner.backbone_ahead)
runner.drain()
