"""Which Python lines launch the expensive torch kernels of a training step (torch.profiler with stacks).
usage (GPU box): python profiles/train_torch_profile.py [epoch]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bench  # noqa: E402
import train_microbench as tm  # noqa: E402
from panopticsegforlargescalepointcloud_amd.training import train_step  # noqa: E402


def main():
    epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda", 0)
    scene, tiles, _ = bench.build_scene(80_000 * 4, 2, 0.05, 2022)
    model = bench.build_model(dev, 0.05)[0].train()
    data, n = tm.make_batch(scene, tiles, [0, 1, 2, 3])
    data = data.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for _ in range(3):
        train_step(model, data, opt, epoch, dev, 1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_step(model, data, opt, epoch, dev, 1)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60,
                                                      max_src_column_width=110))


if __name__ == "__main__":
    main()
