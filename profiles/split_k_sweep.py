"""Split-K sweep on the small (deep) levels of the bench scene: pp_spconv_fwd_ex with split_k = auto / 1 / 2 / 4 / 8.
usage (GPU box): python profiles/split_k_sweep.py <n_tiles>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
dev = torch.device("cuda")
coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
cm = ME.CoordinateManager(coords)
ts = 1
for want, c in [(8, 64), (16, 80), (32, 96), (64, 112)]:
    while ts < want:
        ts = cm.ensure_stride(ts, 2)
    nbr, n = cm.kernel_map(ts, ts, 3, 1), cm.level(ts).n
    x = torch.randn(n, c, device=dev)
    pk = ops.pack_weight(torch.randn(27, c, c, device=dev) * 0.05)
    line = "ts=%d rows %d %d->%d:" % (ts, n, c, c)
    for split in (0, 1, 2, 4, 8):
        fn = lambda: ops.spconv_fwd(x, pk, nbr, n, c, 27, variant=(0, 0, split))  # noqa: E731
        try:
            fn()
        except Exception:
            line += "  split %s n/a" % split
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        line += "  split %s %.0f us" % ("auto" if split == 0 else split, e0.elapsed_time(e1) * 100)
    print(line)
