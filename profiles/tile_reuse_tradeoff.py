"""Tile coherence vs row reuse of a same-level kernel map under three row orders: Morton blocks (PP_MAP_ORDER=0), mask-sorted
inside every 256-row group (what an LDS row cache per workgroup could afford), mask-sorted inside 8192-row windows (product).
Prints executed tile rows per useful pair (MFMA waste) and distinct input rows per 256-row group (what a row cache must hold).
usage (GPU box): PP_MAP_ORDER=0 python profiles/tile_reuse_tradeoff.py [n_tiles]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, synthetic as syn  # noqa: E402

assert not ME.MAP_ORDER, "run with PP_MAP_ORDER=0: the orders are applied here"
n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scene, tiles, _ = bench.build_scene(160_000 * n_tiles, int(np.ceil(np.sqrt(n_tiles))), 0.05, 2022)
b = syn.tile_batch(scene, tiles, list(range(min(n_tiles, len(tiles)))))
dev = torch.device("cuda")
coords = torch.cat([torch.from_numpy(b["batch"]).int()[:, None], torch.from_numpy(b["coords"]).int()], 1).to(dev)
nbr = ME.CoordinateManager(coords).kernel_map(1, 1, 3, 1)  # Morton block order
K, n = nbr.shape
present = nbr >= 0
mask = (present.long() << torch.arange(K, device=dev)[:, None]).sum(0)
pairs = int(present.sum())


def report(name, order):
    """order: new row -> old row (rows AND neighbour ids are renumbered, as the slot order does)"""
    inv = torch.empty_like(order)
    inv[order] = torch.arange(n, device=dev)
    m = nbr[:, order]
    m = torch.where(m >= 0, inv[m.clamp(min=0)], m)
    pad = (-n) % 16
    p = torch.nn.functional.pad(m >= 0, (0, pad)).reshape(K, -1, 16).any(2)
    executed = int(p.sum()) * 16
    G = (n + 255) // 256
    grp = (torch.arange(n, device=dev) // 256).repeat(K, 1)
    key = (grp.long() * (n + 1) + m.long())[m >= 0]
    cnt = torch.bincount(torch.unique(key) // (n + 1), minlength=G).float()
    print("%-34s executed tile rows / useful pair %.2f   distinct rows per 256-row group: mean %.0f p90 %.0f max %.0f" % (
        name, executed / pairs, cnt.mean(), cnt.quantile(0.9), cnt.max()))


ident = torch.arange(n, device=dev)
report("Morton blocks", ident)
for W in (256, 1024, 8192, 32768, 131072, 1 << 30):
    key = (ident // W) * (1 << 27) + mask
    report("mask-sorted inside %d-row windows" % W, torch.sort(key, stable=True)[1])

# bit significance inside 8192-row windows: raw offset index (above), the product's class order (centre < faces < edges <
# corners), and a per-window order by frequency (the rarest offset of THAT window most significant)
W = 8192
bits = ((mask[:, None] >> torch.arange(K, device=dev)[None, :]) & 1)  # [n, 27]
corner = [0, 2, 6, 8, 18, 20, 24, 26]
face = [4, 10, 12, 14, 16, 22]
edge = [k for k in range(27) if k not in corner + face + [13]]
cls = [13] + face + edge + corner
key = (ident // W) * (1 << 27) + (bits[:, cls] << torch.arange(K, device=dev)[None, :]).sum(1)
report("class order (product), 8192-row windows", torch.sort(key, stable=True)[1])
nw = (n + W - 1) // W
freq = torch.zeros(nw, K, device=dev).index_add_(0, ident // W, bits.float())
for name, score in [("per-window frequency order (rarest highest)", -freq),
                    ("per-window order (closest to half of the rows highest)", -(freq - W / 2).abs())]:
    rank = torch.argsort(torch.argsort(score, dim=1, descending=False), dim=1)  # position of bit k in the key: low score -> low position
    pos = rank[ident // W]                                                      # [n, 27]
    key = (ident // W) * (1 << 27) + (bits << pos).sum(1)
    report(name, torch.sort(key, stable=True)[1])

# one more level of the same idea: after the 8192-row sort, every 1024-row chunk of the result is sorted again by ITS OWN
# frequency order (a two-level approximation of a per-window decision tree)
rank = torch.argsort(torch.argsort(-freq, dim=1), dim=1)
key = (ident // W) * (1 << 27) + (bits << rank[ident // W]).sum(1)
o1 = torch.sort(key, stable=True)[1]
for W2 in (2048, 1024, 256):
    b2 = bits[o1]
    f2 = torch.zeros((n + W2 - 1) // W2, K, device=dev).index_add_(0, ident // W2, b2.float())
    r2 = torch.argsort(torch.argsort(-f2, dim=1), dim=1)
    k2 = (ident // W2) * (1 << 27) + (b2 << r2[ident // W2]).sum(1)
    report("frequency order, 8192 then %d-row chunks" % W2, o1[torch.sort(k2, stable=True)[1]])
