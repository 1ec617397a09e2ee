"""Convolution shapes on the kernel maps of the bench scene (slot-ordered, as the model uses them), for timing, PMC
passes and A/B builds of the library (PP_HIP_LIB).
usage (GPU box): python profiles/conv_one.py <n_tiles> <shape>[,<shape>...] [reps]
   shape = ts:cin:cout[:kind]   kind = same (default) | up (transposed, ts*2 -> ts) | down (strided, ts/2 -> ts)
   e.g.  python profiles/conv_one.py 16 1:16:16,2:32:32,4:48:48,1:64:64:up
   PP_CONV_VARIANT="rows_per_wave,pipeline,split_k" selects an explicit kernel variant (pp_spconv_fwd_ex)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402


def main():
    n_tiles = int(sys.argv[1])
    shapes = sys.argv[2].split(",")
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    cm = ME.CoordinateManager(coords)
    total = 0.0
    for shape in shapes:
        f = shape.split(":")
        ts_want, cin, cout = int(f[0]), int(f[1]), int(f[2])
        kind = f[3] if len(f) > 3 else "same"
        ts = 1
        top = ts_want * 2 if kind == "up" else ts_want
        while ts < top:
            ts = cm.ensure_stride(ts, 2)
        if kind == "same":
            nbr, n_in = cm.kernel_map(ts_want, ts_want, 3, 1), cm.level(ts_want).n
        elif kind == "up":
            cm.kernel_map(ts_want, ts_want * 2, 3, 1)
            nbr, n_in = cm.kernel_map(ts_want * 2, ts_want, 3, -1), cm.level(ts_want * 2).n
        else:
            nbr, n_in = cm.kernel_map(ts_want // 2, ts_want, 3, 1), cm.level(ts_want // 2).n
        n = cm.level(ts_want).n
        order = getattr(nbr, "pp_order", None)
        P = int(ops._pairs_of(nbr).item())
        x = torch.randn(n_in, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        pk = ops.pack_weight(w)
        var = tuple(int(v) for v in os.environ["PP_CONV_VARIANT"].split(",")) if os.environ.get("PP_CONV_VARIANT") else None
        fn = lambda: ops.spconv_fwd(x, pk, nbr, n, cout, 27, row_order=order, variant=var)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        total += us
        print("ts=%d %s rows %d pairs/row %.2f  %d->%d: %.1f us  %.1f TF useful" % (ts_want, kind, n, P / n, cin, cout, us,
                                                                               2.0 * P * cin * cout / us / 1e6))
    print("TOTAL %.1f us" % total)


if __name__ == "__main__":
    main()
