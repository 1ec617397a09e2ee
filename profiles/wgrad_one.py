"""Weight-gradient launches on the kernel maps of a training-sized batch (slot-ordered, as the model uses them) and on two
synthetic maps that bracket them: "dense" (every neighbour present, nbr[k][i] = i: perfect locality) and "shuffled"
(every neighbour present, random rows).  For timing the dense-map kernel (k_spconv_bww2) against the pair-major one (k_spconv_bww4).
usage (GPU box): python profiles/wgrad_one.py <shape>[,<shape>...] [reps]      shape = ts:cin:cout"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    shapes = sys.argv[1].split(",")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    scene, tiles, _ = bench.build_scene(80_000 * 4, 2, 0.05, 2022)
    b = syn.tile_batch(scene, tiles, [0, 1, 2, 3])
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    cm = ME.CoordinateManager(coords)
    for shape in shapes:
        ts_want, cin, cout = (int(v) for v in shape.split(":"))
        ts = 1
        while ts < ts_want:
            ts = cm.ensure_stride(ts, 2)
        nbr = cm.kernel_map(ts_want, ts_want, 3, 1)
        n = cm.level(ts_want).n
        P = int(ops._pairs_of(nbr).item())
        x = torch.randn(n, cin, device=dev)
        g = torch.randn(n, cout, device=dev)
        dense = torch.arange(n, device=dev, dtype=torch.int32).repeat(27, 1).contiguous()
        shuf = torch.stack([torch.randperm(n, device=dev).int() for _ in range(27)]).contiguous()
        maps = [("model map", nbr, P), ("dense", dense, 27 * n), ("shuffled", shuf, 27 * n)]
        if os.environ.get("WGRAD_MAPS"):   # e.g. WGRAD_MAPS=model: only that map (counter passes)
            maps = [m for m in maps if m[0].split()[0] in os.environ["WGRAD_MAPS"].split(",")]
        for name, m, pairs in maps:
            us = timed(lambda: ops.spconv_bwd_weight(x, g, m, 27), reps)
            t_build = timed(lambda: ops.wgrad_pairs(m, 27), reps)
            wp = ops.wgrad_pairs(m, 27)
            us_p = timed(lambda: ops.spconv_bwd_weight_pairs(x, g, wp), reps)
            tf = 2.0 * pairs * cin * cout / 1e6
            print("ts=%d rows %d %-9s pairs/row %5.2f  %3d->%3d: dense map %7.1f us %5.1f TF (%.3f) | pair lists %7.1f us %5.1f TF "
                  "(%.3f of 157.3), lists built in %.1f us" % (ts_want, n, name, pairs / n, cin, cout, us, tf / us, tf / us / 157.3,
                                                              us_p, tf / us_p, tf / us_p / 157.3, t_build))


if __name__ == "__main__":
    main()
