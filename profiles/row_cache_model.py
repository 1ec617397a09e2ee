"""CPU model of the "LDS row cache" convolution VERDICT r02 asked for (no GPU needed; oracle/ builds the kernel maps).
For every level of the bench scene (first N tiles) and several row orders it prints
  * executed tile rows per useful pair (MFMA waste of the output-stationary kernel) for mask-sort windows of W rows,
  * per group of G consecutive output rows: the share of pairs whose input row lies in the group itself ("intra"), the
    number of DISTINCT input rows per output row (what a per-workgroup row cache loads instead of one gather per pair),
    the 99th percentile / maximum of distinct rows per group (what the cache must hold), and the share of executed
    16-row tile loads that would still need a global gather if only the group's own rows were cached.
usage: python profiles/row_cache_model.py [n_tiles=4] > profiles/r03_row_cache_model.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle  # noqa: E402
from panopticsegforlargescalepointcloud_amd import synthetic as syn  # noqa: E402

K = 27


def spread3(x):
    x = x.astype(np.uint64) & np.uint64(0xFFFF)
    for sh, m in ((32, 0x1F00000000FFFF), (16, 0x1F0000FF0000FF), (8, 0x100F00F00F00F00F), (4, 0x10C30C30C30C30C3), (2, 0x1249249249249249)):
        x = (x | (x << np.uint64(sh))) & np.uint64(m)
    return x


def order_key(c, ts, bb=4):
    """the product's block order (csrc/pp_common.h pp_order_key, block_bits 4)"""
    sh, hb = int(np.log2(ts)), bb - 1
    key = c[:, 0].astype(np.uint64) << np.uint64(48)
    for axis in range(3):
        q = ((c[:, axis + 1] + 32768) >> sh).astype(np.uint32)
        key |= ((spread3(q >> bb) << np.uint64(3 * bb)) | ((q & 1).astype(np.uint64) << np.uint64(3 * hb)) |
                spread3((q >> 1) & ((1 << hb) - 1))) << np.uint64(axis)
    return key


def analyse(coords, ts):
    coords = coords[np.argsort(order_key(coords, ts), kind="stable")]
    n = len(coords)
    nbr = oracle.kernel_map(coords, coords, 3, ts, 1)
    present = nbr >= 0
    pairs = int(present.sum())
    mask = (present.astype(np.int64) << np.arange(K)[:, None]).sum(0)
    bits = (mask[:, None] >> np.arange(K)[None, :]) & 1
    ident = np.arange(n)
    print("== tensor stride %d: %d rows, %.2f pairs/row" % (ts, n, pairs / n), flush=True)

    def report(name, order, groups):
        inv = np.empty(n, np.int64)
        inv[order] = ident
        m = nbr[:, order]
        m = np.where(m >= 0, inv[np.maximum(m, 0)], -1)
        pad = (-n) % 16
        p = np.pad(m >= 0, ((0, 0), (0, pad))).reshape(K, -1, 16).any(2)
        out = "%-26s executed/useful %.2f" % (name, int(p.sum()) * 16 / pairs)
        for G in groups:
            grp = ident // G
            intra = ((m // G) == grp[None, :]) & (m >= 0)
            keyu = (np.broadcast_to(grp[None, :], m.shape).astype(np.int64) * (n + 1) + m)[m >= 0]
            cnt = np.bincount(np.unique(keyu) // (n + 1), minlength=(n + G - 1) // G)
            extra = (m >= 0) & ~intra
            pe = np.pad(extra, ((0, 0), (0, pad))).reshape(K, -1, 16).any(2)
            out += " | G=%d intra %.2f distinct/row %.2f p99 %d max %d loads-needing-global %.2f" % (
                G, intra.sum() / pairs, cnt.mean() / G, np.quantile(cnt, 0.99), cnt.max(), pe.sum() / max(p.sum(), 1))
        print(out, flush=True)

    def freq_order(W):  # the product's order: rows sorted by mask inside windows of W rows, rarest offset of the window first
        nw = (n + W - 1) // W
        freq = np.zeros((nw, K))
        np.add.at(freq, ident // W, bits)
        rank = np.argsort(np.argsort(-freq, axis=1, kind="stable"), axis=1, kind="stable")
        k = (ident // W).astype(np.int64) * (1 << 27) + (bits << rank[ident // W]).sum(1)
        return np.argsort(k, kind="stable")

    report("block order", ident, (256, 512, 1024))
    for W in (256, 512, 1024, 8192, 32768, 131072, 1 << 30):
        report("mask-sorted, W=%d" % W, freq_order(W), (256, 512, 1024) if W <= 1024 else (256,))


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    t0 = time.time()
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    c = np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)
    analyse(c, 1)
    for ts in (2, 4, 8):
        c = np.unique(np.concatenate([c[:, :1], (c[:, 1:] // ts) * ts], 1), axis=0).astype(np.int32)
        analyse(c, ts)
    print("(%.0f s on the host)" % (time.time() - t0))


main()
