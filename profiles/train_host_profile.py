"""Host-side cost centres of a training step (cProfile over 5 steps, GPU work asynchronous): which Python functions the
host thread spends the step in.  usage (GPU box): python profiles/train_host_profile.py [epoch]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bench  # noqa: E402
import train_microbench as tm  # noqa: E402
from panopticsegforlargescalepointcloud_amd.training import train_step  # noqa: E402


def main():
    epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda", 0)
    scene, tiles, _ = bench.build_scene(80_000 * 4, 2, 0.05, 2022)
    model = bench.build_model(dev, 0.05)[0].train()
    data, n = tm.make_batch(scene, tiles, [0, 1, 2, 3])
    data = data.to(dev)
    fused = os.environ.get("PP_ADAM", "fused") == "fused"   # torch's single-launch Adam; "foreach" = torch's default on a GPU
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=fused)
    for _ in range(3):
        train_step(model, data, opt, epoch, dev, 1)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        train_step(model, data, opt, epoch, dev, 1)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumtime").print_stats(60)


if __name__ == "__main__":
    main()
