"""End-to-end throughput on the other single-GPU configurations of BASELINE.json (the bench metric is configs[3]):
  C2  NPM3D-like full scene, ~2 M points, 5 cm voxels        (bench.py --points 2000000 --grid 4)
  C3  FOR-instance-like forest tile set, ~1 M points, 10 cm voxels, r = 8 m cylinders, two classes, offset + embedding
      dual clustering (cluster_type of the published setting), same timed region as bench.py.
usage (GPU box): python profiles/config_microbench.py [--out DIR]      (DIR/c2.json, DIR/c3.json: one JSON object each)"""
import copy
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import panoptic, synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.scene import TileRunner  # noqa: E402


OUT = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None


def _save(name, obj):
    if OUT:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f)


def c2():
    """the bench's own JSON line for configs[1] (incl. roofline and its self-check: batch invariance + oracle parity of the
    median tile, which needs the CPU pass -- a bounded one)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", "2000000", "--grid", "4", "--steps", "10",
                          "--warmup", "2"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    d["config"]["baseline_config"] = "BASELINE.json configs[1]: NPM3D full scene (~2M pts, 5 cm voxel) on 1 x MI355X"
    _save("c2.json", d)
    print("C2  %s: %.1f ms/step, %.1f M points/s, checks %s" % (d["config"]["workload"][:90], d["ms_per_step"], d["value"] / 1e6,
                                                                d["config"].get("checks", {}).get("all")))


def c3(n_points=1_000_000, voxel=0.10, grid=10, steps=5):
    dev = torch.device("cuda", 0)

    class DS:
        feature_dimension = 4
        num_classes = syn.FOR_NUM_CLASSES
        stuff_classes = torch.tensor(syn.FOR_STUFF)
    _, cfg, _ = bench.build_model(dev, voxel)
    cfg = copy.deepcopy(cfg)
    torch.manual_seed(11)
    model = panoptic.PointGroup3heads(cfg, "dummy", DS, None)
    with torch.no_grad():
        model.ScorerHead[0].bias.fill_(1.0)
    model = model.to(dev).eval()
    scene = syn.forest_scene(n_points, voxel, 2022)
    tiles, radius = syn.cylinder_tiles(scene, grid)
    ids = list(range(len(tiles)))
    b = syn.tile_batch(scene, tiles, ids)
    ov = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(6), offset_sigma=0.1)
    dev_b = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    override = tuple(torch.from_numpy(a).to(dev) for a in ov)
    runner = TileRunner(model, dev)
    for _ in range(2):
        labels, res, counts = runner.run(dev_b, len(ids), override=override)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        labels, res, counts = runner.run(dev_b, len(ids), override=override, next_batch=dev_b)  # (a stream of batches, as bench.py)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = len(b["pos"])
    # self-check (untimed): the median tile of the batch run alone gives bit-identical instance labels
    sizes = np.bincount(b["batch"])
    j = int(np.argsort(sizes)[len(sizes) // 2])
    m = torch.from_numpy(b["batch"] == j).to(dev)
    one = {k: (v[m] if k != "batch" else torch.zeros(int(m.sum()), dtype=v.dtype, device=dev)) for k, v in dev_b.items()}
    l1, r1, c1 = runner.run(one, 1, override=tuple(o[m] for o in override))
    ok = bool(torch.equal(labels[m], l1)) and counts[j] == c1[0]
    _save("c3.json", {"metric": "points/sec end-to-end (sparse-conv fwd + clustering)", "value": n / dt, "unit": "points/sec",
                      "n_gpus": 1, "steps": steps, "warmup": 2, "ms_per_step": 1e3 * dt, "dtype": "f32", "data": "synthetic",
                      "config": {"baseline_config": "BASELINE.json configs[2]: FOR-instance forest tile (r = 8 m cylinders, ~1M pts, 10 cm "
                                                    "voxel), offset + embedding dual clustering on 1 x MI355X",
                                 "workload": "synthetic forest, %d cylinders (r = %.1f m), %d voxels fed (%d scene voxels, %d trees), "
                                             "cluster_type %s" % (len(ids), radius, n, len(scene.pos), scene.n_inst, cfg.cluster_type),
                                 "proposals_per_step": int(res.clusters_csr.n), "instances_per_step": int(sum(counts)),
                                 "checks": {"batch_invariance": "pass" if ok else "FAIL", "oracle": "tests/test_forest_gpu.py (same model "
                                            "and generator at test size)", "all": "pass" if ok else "FAIL"}}})
    print("C3  forest: %d cylinders (r = %.1f m), %d voxels fed (%d scene voxels, %d trees), cluster_type %s: %.1f ms/step, "
          "%.1f M points/s, %d proposals -> %d instances" % (len(ids), radius, n, len(scene.pos), scene.n_inst, cfg.cluster_type,
                                                               1e3 * dt, n / dt / 1e6, res.clusters_csr.n, sum(counts)))


if __name__ == "__main__":
    if "--c3-only" not in sys.argv:
        c2()
    c3()
