"""The library's radix sort / scan alone: 64-bit keys of the bench scene's size (Morton ordering of 10 M rows).
usage (GPU box): python profiles/sort_one.py [n] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator(device="cuda").manual_seed(1)
for dtype, bits in [(torch.int64, 64), (torch.int32, 32), (torch.int32, 20)]:
    keys = torch.randint(0, 2 ** 31 - 1 if dtype == torch.int32 else 2 ** 62, (n,), dtype=dtype, device="cuda", generator=g)
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ops.sort_pairs(keys, vals, bits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.sort_pairs(keys, vals, bits)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps):
        torch.sort(keys, stable=True)
    e1.record()
    torch.cuda.synchronize()
    print("%d keys of %d bits (%s): sort_pairs %.3f ms (%.0f us per 8-bit pass), torch.sort %.3f ms" % (n, bits, dtype, t, 1e3 * t / ((bits + 7) // 8), e0.elapsed_time(e1) / reps))
x = torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda", generator=g)
ops.exclusive_scan(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.exclusive_scan(x)
e1.record()
torch.cuda.synchronize()
print("exclusive_scan of %d int32: %.1f us" % (n, 1e3 * e0.elapsed_time(e1) / reps))
