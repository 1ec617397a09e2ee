"""Which lines of the training step synchronise with the GPU: one step under torch.cuda.set_sync_debug_mode("warn"), warnings
grouped by the first frame inside this package.  (Reads issued from C -- region growing, mean shift, block index -- are not
seen by torch.)  usage (GPU box): python profiles/train_sync_trace.py [epoch=1]"""
import collections
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bench  # noqa: E402
import train_microbench as tm  # noqa: E402
from panopticsegforlargescalepointcloud_amd.training import train_step  # noqa: E402

epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
scene, tiles, _ = bench.build_scene(80_000 * 4, 2, 0.05, 2022)
model = bench.build_model(dev, 0.05)[0].train()
data, n = tm.make_batch(scene, tiles, [0, 1, 2, 3])
data = data.to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
for _ in range(3):
    train_step(model, data, opt, epoch, dev, 1)
torch.cuda.synchronize()
sites = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    for fr in reversed(traceback.extract_stack()):
        if "panopticsegforlargescalepointcloud_amd" in fr.filename:
            sites[(os.path.basename(fr.filename), fr.lineno, (fr.line or "").strip()[:90])] += 1
            return
    sites[("?", 0, str(message)[:60])] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
train_step(model, data, opt, epoch, dev, 1)
torch.cuda.set_sync_debug_mode("default")
print("%d synchronising torch calls in one training step (epoch %d, %d voxels)" % (sum(sites.values()), epoch, n))
for (f, l, src), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%3d  %s:%d  %s" % (c, f, l, src))
