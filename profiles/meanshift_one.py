"""Mean shift on the embeddings of the bench scene (synthetic head statistics), stand-alone: time per call of the model's
_embed_clusters path (mask -> rows -> pp_meanshift -> group_by_key).  Run under rocprofv3 --kernel-trace --stats for the
per-kernel split.   usage (GPU box): python profiles/meanshift_one.py <n_tiles> [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import ops, synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.utils import meanshift_cluster  # noqa: E402


def main():
    n_tiles = int(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    rng = np.random.default_rng(2022)
    cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, rng)
    dev = torch.device("cuda")
    pred = torch.from_numpy(cls).to(dev)
    emb = torch.from_numpy(emb).to(dev)
    batch = torch.from_numpy(b["batch"]).to(dev)
    stuff = torch.tensor(syn.NPM3D_STUFF, device=dev)
    model, cfg, DS = bench.build_model(dev, 0.05)

    def fn():
        mask = ops.not_ignored(pred, stuff, syn.NPM3D_NUM_CLASSES)
        ind = torch.nonzero(mask).view(-1)
        return meanshift_cluster.cluster_single_csr(emb[ind], batch[ind], ind, cfg.bandwidth)

    csr = fn()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("points %d  clusters %d  mean shift %.2f ms per call (wall)" % (pred.shape[0], csr.n, 1e3 * (time.perf_counter() - t0) / reps))


main()
