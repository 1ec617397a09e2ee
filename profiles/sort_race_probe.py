"""k_window_sort<W> beside another stream: which windows fail, beside which kernels, and what the bad entries look like.
usage (GPU box): python profiles/sort_race_probe.py"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bruteforce as bf  # noqa: E402
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402


def main():
    rng = np.random.default_rng(61)
    fine = bf.surface_coords(rng, n_batch=4, n=400000, extent=900)
    d = torch.from_numpy(fine).cuda()
    fine = fine[ops.morton_order(d, 1, 4).cpu().numpy()]
    d = torch.from_numpy(fine).cuda()
    n = len(fine)
    idx, _ = ops.block_index_build(d, 1, 4)
    nbr = ops.kernel_map_bi(d, idx, 3, 1, 1, want_mask=True)
    mask = nbr.pp_mask
    cidx, cc = ops.block_index_coarsen(idx, n)
    m2 = ops.kernel_map_bi(cc, cidx, 3, 2, 1, want_mask=True)
    o2 = ops.map_order(m2.pp_mask, window=16384)
    c2, p2 = ops.level_permute(cc, o2)
    torch.cuda.synchronize()
    noises = {
        "none": lambda: None,
        "coarsen": lambda: ops.block_index_coarsen(idx, n),
        "lookup": lambda: ops.kernel_map_bi(cc, cidx, 3, 2, 1, want_mask=True),
        "sort16384": lambda: ops.map_order(m2.pp_mask, window=16384),
        "sort8192": lambda: ops.map_order(m2.pp_mask, window=8192),
        "permute16384": lambda: ops.map_permute(m2, o2, translate=p2),
        "level_permute": lambda: ops.level_permute(cc, o2),
        "torch_fill": lambda: torch.empty(50_000_000, device="cuda").fill_(1.0),
    }
    ar = torch.arange(n, device="cuda")
    only = os.environ.get("PROBE_NOISES")
    windows = tuple(int(w) for w in os.environ.get("PROBE_WINDOWS", "2048,4096,8192").split(","))
    for name, fn in noises.items():
        if only and name not in only.split(","):
            continue
        for window in windows:
            stop = threading.Event()
            side = torch.cuda.Stream()

            def noise():
                with torch.cuda.stream(side):
                    while not stop.is_set():
                        fn()
                        if name == "none":
                            stop.wait(0.01)

            th = threading.Thread(target=noise)
            th.start()
            bad_iters, detail = 0, ""
            for it in range(40):
                order = ops.map_order(mask, window=window)
                o = order.long()
                bad = torch.nonzero((o // window) != (ar // window)).view(-1)
                perm_ok = bool(torch.equal(torch.sort(o)[0], ar))
                if bad.numel() or not perm_ok:
                    bad_iters += 1
                    if not detail and bad.numel():
                        b = bad[:6]
                        detail = " first bad slots %s values-base %s (window starts %s) permutation %s n_bad %d" % (
                            b.tolist(), (o[b] - (b // window) * window).tolist(), ((b // window) * window).tolist(), perm_ok, bad.numel())
            stop.set()
            th.join()
            torch.cuda.synchronize()
            print("noise %-14s window %5d: %2d / 40 iterations bad%s" % (name, window, bad_iters, detail), flush=True)


if __name__ == "__main__":
    main()
