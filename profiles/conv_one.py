"""One convolution shape on a realistic kernel map, for PMC passes.
usage (GPU box): python profiles/conv_one.py <n_tiles> <ts> <cin> <cout> <dense|rb> [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402


def main():
    n_tiles, ts_want, cin, cout, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    cm = ME.CoordinateManager(coords)
    ts = 1
    while ts < ts_want:
        ts = cm.ensure_stride(ts, 2)
    n = cm.level(ts).n
    nbr = cm.kernel_map(ts, ts, 3, 1)
    P = int((nbr >= 0).sum().item())
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    pk = ops.pack_weight(w)
    if mode == "rb":
        rb = cm.rulebook(ts, ts, 3, 1)
        fn = lambda: ops.spconv_fwd_rb(x, pk, rb, cout)  # noqa: E731
    else:
        fn = lambda: ops.spconv_fwd(x, pk, nbr, n, cout, 27)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("ts=%d rows %d pairs/row %.2f  %d->%d %s: %.1f us  %.1f TF useful" % (ts, n, P / n, cin, cout, mode, us,
                                                                             2.0 * P * cin * cout / us / 1e6))


if __name__ == "__main__":
    main()
