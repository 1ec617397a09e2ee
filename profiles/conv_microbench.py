"""Micro-benchmark of the two sparse-conv kernels on realistic kernel maps (synthetic urban tiles).
usage (GPU box): python profiles/conv_microbench.py [n_tiles] [only=rb|dense]
Prints per (level, Cin, Cout): pairs/row, time of the dense-offset kernel, time of the rulebook kernel, useful TFLOP/s."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    only = sys.argv[2] if len(sys.argv) > 2 else "both"
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    cm = ME.CoordinateManager(coords)
    ts = 1
    shapes = {1: [(16, 16), (64, 64), (32, 16)], 2: [(32, 32), (96, 96)], 4: [(48, 48), (128, 128)], 8: [(64, 64)], 16: [(80, 80)]}
    print("rows per level / maps built on", coords.shape[0], "input rows")
    for lvl in range(5):
        n = cm.level(ts).n
        nbr = cm.kernel_map(ts, ts, 3, 1)
        P = int((nbr >= 0).sum().item())
        rb = cm.rulebook(ts, ts, 3, 1)
        print("level ts=%d rows %d pairs/row %.2f rulebook entries/pairs %.3f" % (ts, n, P / n, rb.total / max(P, 1)))
        for cin, cout in shapes[ts]:
            x = torch.randn(n, cin, device=dev)
            w = torch.randn(27, cin, cout, device=dev) * 0.05
            pk = ops.pack_weight(w)
            td = tr = float("nan")
            if only in ("both", "dense"):
                td = timeit(lambda: ops.spconv_fwd(x, pk, nbr, n, cout, 27))
            if only in ("both", "rb"):
                tr = timeit(lambda: ops.spconv_fwd_rb(x, pk, rb, cout))
            fl = 2.0 * P * cin * cout
            print("   %3d->%3d  dense %8.1f us (%5.1f TF useful)   rulebook %8.1f us (%5.1f TF useful)" %
                  (cin, cout, td, fl / td / 1e6, tr, fl / tr / 1e6))
        # transposed strided map onto this level from the next
        ts2 = cm.ensure_stride(ts, 2)
        down = cm.kernel_map(ts, ts2, 3, 1)
        up = cm.kernel_map(ts2, ts, 3, -1)
        Pu = int((up >= 0).sum().item())
        rbu = cm.rulebook(ts2, ts, 3, -1)
        nc = cm.level(ts2).n
        cin, cout = {1: (64, 64), 2: (96, 96), 4: (128, 128), 8: (160, 160), 16: (192, 192)}[ts]
        x = torch.randn(nc, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        pk = ops.pack_weight(w)
        td = tr = float("nan")
        if only in ("both", "dense"):
            td = timeit(lambda: ops.spconv_fwd(x, pk, up, n, cout, 27))
        if only in ("both", "rb"):
            tr = timeit(lambda: ops.spconv_fwd_rb(x, pk, rbu, cout))
        fl = 2.0 * Pu * cin * cout
        print("   up %3d->%3d (pairs/row %.2f) dense %8.1f us (%5.1f TF)   rulebook %8.1f us (%5.1f TF)" %
              (cin, cout, Pu / n, td, fl / td / 1e6, tr, fl / tr / 1e6))
        ts = ts2


if __name__ == "__main__":
    main()
