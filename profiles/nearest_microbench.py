"""Full-resolution back-projection micro-benchmark (row f2): exact 1-NN of a 30 M-point raw cloud among the 10 M voxel
centres the network saw (`pp_nearest`), plus scipy's cKDTree on a bounded sample of the same queries as the CPU baseline.
usage (GPU box): python profiles/nearest_microbench.py [n_ref] [n_query]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402


def main():
    n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000_000
    dev = torch.device("cuda", 0)
    scene, _, _ = bench.build_scene(n_ref, 64, 0.05, 2022)
    ref = torch.from_numpy(scene.pos.astype(np.float32)).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    pick = torch.randint(0, ref.shape[0], (n_q,), device=dev, generator=g)
    q = ref[pick] + (torch.rand((n_q, 3), device=dev, generator=g) - 0.5) * 0.05  # raw points inside their voxel
    for cell in (0.1, 0.15, 0.25):
        ops.nearest(ref, q[:1000], cell)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, d2 = ops.nearest(ref, q, cell)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print("pp_nearest  %d refs, %d queries, cell %.2f: %.1f ms  (%.1f M queries/s, %.1f %% own voxel)" %
              (ref.shape[0], n_q, cell, 1e3 * t, n_q / t / 1e6, 100.0 * float((idx == pick).float().mean())))
    from scipy.spatial import cKDTree
    refh = ref.cpu().numpy()
    qs = q[:2_000_000].cpu().numpy()
    t0 = time.perf_counter()
    tree = cKDTree(refh)
    tb = time.perf_counter() - t0
    t0 = time.perf_counter()
    dd, ii = tree.query(qs, k=1, workers=-1)
    tq = time.perf_counter() - t0
    same = float((torch.from_numpy(ii).to(dev) == idx[:len(qs)]).float().mean())
    print("cKDTree     build %.1f s, %d queries in %.1f s on %d cores (%.2f M queries/s); same neighbour %.4f" %
          (tb, len(qs), tq, os.cpu_count(), len(qs) / tq / 1e6, same))


if __name__ == "__main__":
    main()
