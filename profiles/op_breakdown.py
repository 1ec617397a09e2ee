"""Per-operation wall time of one end-to-end step (synchronising around every call of the `ops` front-end, so the sum is
larger than the pipelined step; what matters is the ranking).  usage (GPU box): python profiles/op_breakdown.py [points] [grid]"""
import collections
import os
import sys
import time

# serial order, one stream: the overlapped side-stream stages would otherwise run inside other calls' timed windows
os.environ["PP_MAP_PREFETCH"] = "0"
os.environ["PP_CLUSTER_OVERLAP"] = "0"

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402

ACC = collections.defaultdict(list)
DEPTH = [0]


def wrap(name):
    fn = getattr(ops, name)

    def timed(*a, **k):
        if DEPTH[0]:
            return fn(*a, **k)
        DEPTH[0] += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            ACC[name].append(time.perf_counter() - t0)
            DEPTH[0] -= 1
    setattr(ops, name, timed)


NAMES = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n[0].islower()
         and getattr(getattr(ops, n), "__module__", "") == ops.__name__]
for n in NAMES:
    wrap(n)

STEPS = 3  # bench.py's 2 priming passes + 1 timed step; only the calls of the last step are reported
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--points", sys.argv[1] if len(sys.argv) > 1 else "10000000",
            "--grid", sys.argv[2] if len(sys.argv) > 2 else "8"]
import runpy  # noqa: E402

_orig = None
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
last = {n: v[len(v) - len(v) // STEPS:] for n, v in ACC.items() if len(v) >= STEPS}
tot = sum(sum(v) for v in last.values())
print("\nops front-end calls of the last step, synchronised around every call: %.1f ms" % (1e3 * tot))
for name, v in sorted(last.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print("  %-28s %4d calls  %8.2f ms" % (name, len(v), 1e3 * sum(v)))
