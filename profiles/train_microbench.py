"""Training-step micro-benchmark (BASELINE.json configs[4] shape on ONE GPU, fp32 like the reference):
batch of 4 cylinders, forward (train-mode BN through the HIP statistics kernels) + losses + backward through every
sparse convolution + Adam step.  usage (GPU box): python profiles/train_microbench.py [n_cylinders] [voxels_per_cyl]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import synthetic as syn  # noqa: E402
from panopticsegforlargescalepointcloud_amd.applications import Data  # noqa: E402


def make_batch(scene, tiles, ids):
    b = syn.tile_batch(scene, tiles, ids)
    oid = b["origin_id"]
    inst = scene.inst[oid]
    inst_local = np.zeros_like(inst)
    for t in np.unique(b["batch"]):
        m = (b["batch"] == t) & (inst > 0)
        _, inv = np.unique(inst[m], return_inverse=True)
        inst_local[m] = inv + 1
    vote = (scene.inst_center[inst] - scene.pos[oid]).astype(np.float32)
    return Data(pos=torch.from_numpy(b["pos"]), coords=torch.from_numpy(b["coords"]), x=torch.from_numpy(b["x"]),
                batch=torch.from_numpy(b["batch"]), y=torch.from_numpy(scene.cls[oid]),
                instance_labels=torch.from_numpy(inst_local), instance_mask=torch.from_numpy(inst > 0),
                vote_label=torch.from_numpy(vote), center_label=torch.from_numpy(scene.inst_center[inst]),
                num_instances=torch.tensor([int(inst_local.max())])), len(oid)


def main():
    ncyl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 80_000
    dev = torch.device("cuda")
    scene, tiles, _ = bench.build_scene(per * 4, 2, 0.05, 2022)
    model = bench.build_model(dev, 0.05)[0].train()
    data, n = make_batch(scene, tiles, list(range(ncyl)))
    data = data.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for epoch, tag in [(1, "epoch <= prepare_epoch (heads + losses)"), (100, "epoch > prepare_epoch (+ grouping, ScorerUnet, score loss)")]:
        times = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.set_input(data, dev)
            opt.zero_grad(set_to_none=True)
            model.forward(epoch=epoch)
            model.backward(epoch)
            opt.step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = float(np.median(times[2:]))
        print("%-62s %7.1f ms/step  %6.2f M points/s  (batch %d cylinders, %d voxels, loss %.4f)" %
              (tag, 1e3 * t, n / t / 1e6, ncyl, n, float(model.loss)))


if __name__ == "__main__":
    main()
