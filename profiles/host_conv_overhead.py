"""Host cost of one fused convolution layer call (ME.conv_bn_act -> ops.spconv_fwd -> C ABI) on a tiny level, where the
kernel takes ~10 us and the loop is launch-bound: what the Python side of a layer costs per call.
usage (GPU box): python profiles/host_conv_overhead.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops  # noqa: E402

rng = np.random.default_rng(0)
c = np.unique(np.concatenate([np.zeros((3000, 1), np.int32), rng.integers(-20, 20, size=(3000, 3)).astype(np.int32)], 1), axis=0)
x = ME.SparseTensor(torch.randn(len(c), 32).cuda(), torch.from_numpy(c).cuda(), device="cuda")
conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3).cuda().eval()
bn = ME.MinkowskiBatchNorm(32).cuda().eval()
with torch.no_grad():
    for _ in range(50):
        y = ME.conv_bn_act(x, conv, bn, relu=True)
    torch.cuda.synchronize()
    for label, prof in [("profiler off", None), ("profiler on", ops.LaunchProfiler())]:
        ops.PROFILER = prof
        t0 = time.perf_counter()
        for _ in range(2000):
            y = ME.conv_bn_act(x, conv, bn, relu=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: %.1f us per layer call to enqueue, %.1f us incl. the drain" % (label, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
    ops.PROFILER = None
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        y = ME.conv_bn_act(x, conv, bn, relu=True)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
