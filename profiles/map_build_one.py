"""Stand-alone timing of the map-building pipeline on the bench scene: input ordering + block index, lookup, window sort,
level permute, map permute, coarsening, strided / transposed maps -- every stage alone on an idle GPU (HIP events).
usage (GPU box): python profiles/map_build_one.py <n_tiles> [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, ops, synthetic as syn  # noqa: E402


def timed(name, fn, reps, rows, nbytes=None):
    out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    extra = "  %.2f TB/s" % (nbytes / us * 1e-6) if nbytes else ""
    print("%-34s %9.1f us  %6.2f ns/row%s" % (name, us, us * 1e3 / max(rows, 1), extra))
    return out


def main():
    n_tiles = int(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    n = coords.shape[0]
    print("level 0: %d rows" % n)
    perm32, cs = timed("morton_order", lambda: ops.morton_order(coords, 1, ME.ORDER_BLOCK_BITS, want_sorted=True, raw=True), reps, n)
    index, _ = timed("block_index_build", lambda: ops.block_index_build(cs, 1, ME.ORDER_BLOCK_BITS), reps, n)
    nbr = timed("kernel_map_bi (same, +mask)", lambda: ops.kernel_map_bi(cs, index, 3, 1, 1, want_mask=True), reps, n, 124 * n)
    for w in (16384, 32768):  # the level's own order uses larger windows than the cross-level maps (round 4)
        ow = timed("map_order (window %d)" % w, lambda: ops.map_order(nbr.pp_mask, window=w), reps, n)
        _, pw = ops.level_permute(cs, ow)
        timed("map_permute (+translate, %d)" % w, lambda: ops.map_permute(nbr, ow, translate=pw), reps, n, 216 * n)
        timed("map_permute (no translate, %d)" % w, lambda: ops.map_permute(nbr, ow), reps, n, 216 * n)
    order = timed("map_order (window sort)", lambda: ops.map_order(nbr.pp_mask), reps, n)
    coords_p, phys_of = timed("level_permute", lambda: ops.level_permute(cs, order), reps, n)
    same = timed("map_permute (+translate)", lambda: ops.map_permute(nbr, order, translate=phys_of), reps, n, 216 * n)
    timed("map_permute (no translate)", lambda: ops.map_permute(nbr, order), reps, n, 216 * n)
    timed("compose_perm", lambda: ops.compose_perm(perm32, order, n, dev), reps, n)
    cidx, ccoords = timed("block_index_coarsen", lambda: ops.block_index_coarsen(index, n), reps, n)
    nc = ccoords.shape[0]
    print("level 1: %d rows" % nc)
    down = timed("kernel_map_bi (strided, translate)", lambda: ops.kernel_map_bi(ccoords, index, 3, 1, 1, translate=phys_of), reps, nc, 108 * nc)
    up = timed("kernel_map_transpose", lambda: ops.kernel_map_transpose(down, n, order=None), reps, n, 108 * (n + nc))
    mk = timed("map_mask", lambda: ops.map_mask(up), reps, n, 112 * n)
    ou = timed("map_order (transposed)", lambda: ops.map_order(mk), reps, n)
    timed("map_permute (transposed)", lambda: ops.map_permute(up, ou), reps, n, 216 * n)
    # the 8-wide form of the same transposed map (inference path since round 3)
    m8 = timed("kernel_map_transpose8 (+key)", lambda: ops.kernel_map_transpose8(down, n, order=None), reps, n, 108 * nc + 36 * n)
    o8 = timed("map_order (8-wide key)", lambda: ops.map_order(m8[1]), reps, n)
    timed("map_permute (8-wide)", lambda: ops.map_permute(m8[0], o8), reps, n, 64 * n)
    timed("kernel_map_bi (transposed lookup)", lambda: ops.kernel_map_bi(coords_p, cidx, 3, 1, -1, want_mask=True), reps, n, 124 * n)


main()
