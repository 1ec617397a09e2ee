"""Training-step micro-benchmark (BASELINE.json configs[4] shape on ONE GPU, fp32 like the reference):
batch of 4 cylinders, forward (train-mode BN through the HIP statistics kernels) + losses + backward through every
sparse convolution + (N > 1: bucketed gradient all-reduce) + Adam step.
usage (GPU box): python profiles/train_microbench.py [n_cylinders] [voxels_per_cyl]
data parallel:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 profiles/train_microbench.py
                 (every rank trains on its own batch of n_cylinders; PP_DIST_BACKEND=gloo to try it on one GPU)"""
import json
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
from panopticsegforlargescalepointcloud_amd.training import GradientReducer, train_step  # noqa: E402


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
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("PP_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    scene, tiles, _ = bench.build_scene(per * 4 * world, 2 * world, 0.05, 2022)
    model = bench.build_model(dev, 0.05)[0].train()   # same seed on every rank => identical replicas
    ids = [(rank * ncyl + i) % len(tiles) for i in range(ncyl)]
    data, n = make_batch(scene, tiles, ids)
    data = data.to(dev)
    fused = os.environ.get("PP_ADAM", "fused") == "fused"   # torch's single-launch Adam; "foreach" = torch's default on a GPU
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=fused)
    reducer = GradientReducer(model.parameters()) if world > 1 else None  # bucketed all-reduce overlapped with backward
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops
    bf16 = os.environ.get("PP_CONV_DTYPE", "fp32").lower() == "bf16"
    result = {"metric": "training step (fwd + bwd + Adam), BASELINE.json configs[4] shape on one GPU", "unit": "ms/step",
              "dtype": "bf16 convolution compute, fp32 tensors" if bf16 else "f32", "n_gpus": world, "voxels_per_rank": n,
              "cylinders_per_rank": ncyl, "data": "synthetic",
              "optimizer": "torch.optim.Adam(lr=1e-3, %s)" % ("fused=True" if fused else "foreach"), "modes": {}}
    modes = [(1, "epoch <= prepare_epoch (heads + losses)"), (100, "epoch > prepare_epoch (+ grouping, ScorerUnet, score loss)")]
    if os.environ.get("PP_TRAIN_MODES"):  # "1" or "2": one mode only (kernel traces of one mode)
        modes = [modes[int(m) - 1] for m in os.environ["PP_TRAIN_MODES"].split(",")]
    for epoch, tag in modes:
        times = []
        for it in range(7):
            if it == 2:  # one step with per-launch HIP events on the convolution kernels (not part of the median)
                ops.PROFILER = ops.LaunchProfiler()
            elif it == 3:
                prof, ops.PROFILER = ops.PROFILER, None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train_step(model, data, opt, epoch, dev, world, reducer=reducer)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = float(np.median(times[3:7]))
        fwd, dgrad, wgrad = prof.summarize("fwd"), prof.summarize("dgrad"), prof.summarize_wgrad()

        def roof(p):
            """fraction of the roof that bounds the launches: fp32 MFMA 157.3 TFLOP/s for fp32 compute; with bf16 compute
            (one 16x16x16 MFMA per four fp32 ones, tensors still fp32 in memory) the kernels are HBM-bound: 8 TB/s"""
            sec = max(p["ms"], 1e-9) * 1e-3
            tf, gbs = p["flops"] / sec / 1e12, p["bytes"] / sec / 1e9
            bound = "hbm" if bf16 else "mfma"
            return {"launches": p["launches"], "ms": round(p["ms"], 3), "bound": bound, "alg_TFLOPs": round(tf, 2), "alg_GBps": round(gbs, 1),
                    "frac": round(gbs / 8000.0 if bf16 else tf / 157.3, 4), "frac_mfma_fp32": round(tf / 157.3, 4),
                    "frac_hbm": round(gbs / 8000.0, 4)}
        result["modes"][tag] = {"ms_per_step": round(1e3 * t, 2), "points_per_s": round(world * n / t), "loss": float(model.loss.detach()),
                                "roofline": {"k_spconv_fwd3 forward": roof(fwd), "k_spconv_fwd3 input gradient": roof(dgrad),
                                             "k_spconv_bww2 weight gradient": roof(wgrad)}}
        if world > 1:
            # replicas must stay identical: compare a parameter checksum across ranks
            chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
            allc = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(allc, chk)
            assert all(abs(float(c) - float(allc[0])) < 1e-6 * max(1.0, abs(float(allc[0]))) for c in allc), "replicas diverged"
        if rank == 0:
            print("%-62s %7.1f ms/step  %6.2f M points/s  (%d rank(s) x %d cylinders, %d voxels/rank, loss %.4f)" %
                  (tag, 1e3 * t, world * n / t / 1e6, world, ncyl, n, float(model.loss)))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
