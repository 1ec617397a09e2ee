"""Kernel-map statistics of the bench scene: pairs/row and how much of the dense-offset kernel's executed work is useful
at 16-row (MFMA tile) and 32-row (wave) granularity.  usage (GPU box): python profiles/map_stats.py [n_tiles]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, synthetic as syn  # noqa: E402


def vorder_stats(cm, ts, name, W=16384):
    """useful@16 of every map INTO level ts when its rows are processed in (window, parity, same-level mask) order"""
    lvl = cm.level(ts)
    n = lvl.n
    same = cm.kernel_map(ts, ts, 3, 1)
    K = 27
    bits = ((same >= 0).to(torch.int64) << torch.arange(K, device=same.device)[:, None]).sum(0)
    q = (lvl.coords[:, 1:].to(torch.int64) + 32768) // ts
    par = (q[:, 0] & 1) | ((q[:, 1] & 1) << 1) | ((q[:, 2] & 1) << 2)
    key = ((torch.arange(n, device=same.device) // W) << 30) | (par << 27) | bits
    order = torch.sort(key)[1]
    out = [name]
    for label, m in (("same", same), ("up", cm.maps.get((ts * 2, ts, 3, -1))), ("down", cm.maps.get((ts // 2, ts, 3, 1)) if ts > 1 else None)):
        if m is None:
            continue
        v = (m >= 0)
        P = int(v.sum())
        for tag, vv in (("phys", v), ("vord", v[:, order])):
            mm = (n + 15) // 16 * 16
            pad = torch.zeros((K, mm), dtype=torch.bool, device=same.device)
            pad[:, :n] = vv
            act = pad.view(K, mm // 16, 16).any(2)
            out.append("%s %s %.3f" % (label, tag, P / (float(act.sum()) * 16)))
    print("| " + " | ".join(out) + " |")


def stats(nbr, name):
    K, n = nbr.shape
    valid = nbr >= 0
    P = int(valid.sum())
    out = [name, n, "%.2f" % (P / n)]
    for g in (16, 32, 64):
        m = (n + g - 1) // g * g
        v = torch.zeros((K, m), dtype=torch.bool, device=nbr.device)
        v[:, :n] = valid
        act = v.view(K, m // g, g).any(2)
        out.append("%.3f" % (P / (float(act.sum()) * g)))
    # distinct masks per 64-row block (how well would mask-sorting inside a block work?)
    bits = (valid.to(torch.int64) << torch.arange(K, device=nbr.device)[:, None]).sum(0)
    m = (n + 63) // 64 * 64
    b = torch.full((m,), -1, dtype=torch.int64, device=nbr.device)
    b[:n] = bits
    srt = torch.sort(b.view(-1, 64), 1)[0]
    distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
    out.append("%.1f" % float(distinct.float().mean()))
    # ideal: rows sorted by mask inside each 64-row block, then 16-row tiles
    srt_bits = srt.view(-1, 16)
    ok = srt_bits >= 0
    anyk = torch.zeros(srt_bits.shape[0], dtype=torch.int64, device=nbr.device)
    for k in range(K):
        anyk += (((srt_bits >> k) & 1) * ok).any(1).to(torch.int64)
    out.append("%.3f" % (P / (float(anyk.sum()) * 16)))
    # rows re-ordered by mask inside windows of W Morton-consecutive rows (W = 1024 ... all): useful fraction @16
    if name.endswith("same"):
        for W in (1024, 4096, 16384, 1 << 30):
            W = min(W, (n + 15) // 16 * 16)
            m = (n + W - 1) // W * W
            b = torch.full((m,), (1 << 40), dtype=torch.int64, device=nbr.device)
            b[:n] = bits
            srt = torch.sort(b.view(-1, W), 1)[0].reshape(-1, 16)
            ok = srt < (1 << 40)
            anyk = torch.zeros(srt.shape[0], dtype=torch.int64, device=nbr.device)
            for k in range(K):
                anyk += (((srt >> k) & 1).bool() & ok).any(1).to(torch.int64)
            out.append("W%d: %.3f" % (W if W < (1 << 29) else 0, P / (float(anyk.sum()) * 16)))
    print("| " + " | ".join(str(x) for x in out) + " |")


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
    b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
    dev = torch.device("cuda")
    coords = torch.from_numpy(np.concatenate([b["batch"][:, None], b["coords"]], 1).astype(np.int32)).to(dev)
    cm = ME.CoordinateManager(coords)
    print("| map | rows | pairs/row | useful@16 | useful@32 | useful@64 | masks/64 rows | useful@16 mask-sorted in 64 |")
    print("|---|---|---|---|---|---|---|---|")
    ts = 1
    for lvl in range(6):
        stats(cm.kernel_map(ts, ts, 3, 1), "ts%d same" % ts)
        ts2 = cm.ensure_stride(ts, 2)
        stats(cm.kernel_map(ts, ts2, 3, 1), "ts%d->%d down" % (ts, ts2))
        stats(cm.kernel_map(ts2, ts, 3, -1), "ts%d->%d up" % (ts2, ts))
        ts = ts2
    print()
    for ts in (1, 2, 4, 8):
        vorder_stats(cm, ts, "level ts%d" % ts)


if __name__ == "__main__":
    main()
