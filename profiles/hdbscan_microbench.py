"""HDBSCAN kernel timing on synthetic per-tile embeddings (thing points of cylinder tiles), vs the CPU oracle on one tile.
Usage: python profiles/hdbscan_microbench.py [n_tiles] [points_per_tile]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    rng = np.random.default_rng(0)
    xs, offs = [], [0]
    for t in range(n_tiles):
        k = int(rng.integers(20, 60))
        cen = rng.normal(0, 3.0, size=(k, 5))
        ids = np.sort(rng.integers(0, k, size=per))            # instance points are contiguous-ish (voxel order)
        xs.append((cen[ids] + rng.normal(0, 0.15, size=(per, 5))).astype(np.float32))
        offs.append(offs[-1] + per)
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        labels, ncl = ops.hdbscan(x, offs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("gpu hdbscan: %d tiles x %d pts: %.1f ms (%.2f Mpts/s), clusters/tile %.1f" %
              (n_tiles, per, dt * 1e3, n_tiles * per / dt / 1e6, float(ncl.float().mean())))
    if "--cpu" in sys.argv:
        from oracle import oracle
        n = min(per, 6000)
        t0 = time.perf_counter()
        oracle.hdbscan(xs[0][:n], [0, n], count_self=False)
        print("cpu oracle (O(n^2) Prim, 1 core): %d pts %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
        try:
            from sklearn.cluster import HDBSCAN
            t0 = time.perf_counter()
            HDBSCAN(min_cluster_size=15, min_samples=5, cluster_selection_epsilon=0.006).fit_predict(xs[0].astype(np.float64))
            print("sklearn HDBSCAN (kd-tree Prim, 1 core): %d pts %.1f ms" % (per, (time.perf_counter() - t0) * 1e3))
        except Exception as e:  # sklearn may be absent on the GPU box
            print("sklearn unavailable:", e)


if __name__ == "__main__":
    main()
