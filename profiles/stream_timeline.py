"""Per-queue view of a rocprofv3 kernel trace of bench.py: for the last <steps> steps, the busy time and the idle gaps of
the queue that runs the convolutions (the model's stream) and of the side queue (map prefetch), by kernel family.
usage: python profiles/stream_timeline.py <results.db> <steps> <ms_per_step>"""
import sqlite3
import sys
from collections import defaultdict

db, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
con = sqlite3.connect(db)
rows = con.execute("select start, end, name, queue_id, stream_id from kernels order by start").fetchall()
tri = [i for i, r in enumerate(rows) if "k_triad" in r[2]]
rows = rows[: tri[0]] if tri else rows
t1 = max(r[1] for r in rows)
t0 = t1 - steps * ms * 1e6
win = [r for r in rows if r[0] >= t0]


def family(name):
    for key, fam in [("k_spconv", "conv"), ("k_window_sort", "map: sort"), ("k_map_permute", "map: permute"), ("k_kernel_map", "map: lookup/transpose"),
                     ("k_map_mask", "map: mask"), ("k_level_permute", "map: level"), ("k_bic", "level: coarsen"), ("k_bi_", "level: index"),
                     ("k_ball_query", "region grow"), ("k_rg_", "region grow"), ("k_ms_", "mean shift"), ("k_nms", "nms"), ("k_head", "heads"),
                     ("k_gather", "gather"), ("rocprim", "rocprim (sort/scan)"), ("fillBuffer", "fill"), ("copyBuffer", "copy"), ("at::native", "torch"),
                     ("k_gbk", "group_by_key"), ("k_seg", "segment_reduce"), ("k_morton", "morton"), ("k_compose", "map: level")]:
        if key in name:
            return fam
    return name[:40]


by_q = defaultdict(list)
for r in win:
    by_q[r[4]].append(r)
conv_q = max(by_q, key=lambda q: sum(e - s for s, e, n, _, _ in by_q[q] if "k_spconv" in n))
print("window %.1f ms x %d steps; streams: %s (main = %s)" % (ms, steps, {q: len(v) for q, v in by_q.items()}, conv_q))
for q, v in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = defaultdict(float)
    cnt = defaultdict(int)
    gaps = defaultdict(lambda: [0, 0.0])
    prev_end, prev_name = v[0][1], v[0][2]
    for s, e, n, _, _ in v:
        busy[family(n)] += e - s
        cnt[family(n)] += 1
    for s, e, n, _, _ in v[1:]:
        g = s - prev_end
        if g > 5e3:
            gaps[(family(prev_name), family(n))][0] += 1
            gaps[(family(prev_name), family(n))][1] += g
        prev_end, prev_name = max(prev_end, e), n
    tot = sum(busy.values())
    gtot = sum(g for _, g in gaps.values())
    print("\nstream %s%s: busy %.1f ms/step, idle in gaps > 5 us %.1f ms/step, %d dispatches/step" %
          (q, " (main)" if q == conv_q else "", tot / steps / 1e6, gtot / steps / 1e6, len(v) / steps))
    for f, t in sorted(busy.items(), key=lambda kv: -kv[1])[:22]:
        print("   %7.2f ms/step %6.1f x/step  %s" % (t / steps / 1e6, cnt[f] / steps, f))
    if q == conv_q:
        print("  largest idle gaps (after -> before):")
        for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:18]:
            print("   %7.2f ms/step %6.1f x/step  %s -> %s" % (g / steps / 1e6, c / steps, a, b))

# the main stream's individual gaps > 50 us in the last step: what ran before / after and what the other streams finished
# just before the gap closed (the item the main stream was waiting for)
main = [r for r in by_q[conv_q] if r[0] >= t1 - ms * 1e6]
others = [r for q, v in by_q.items() if q != conv_q for r in v]
print("\nmain-stream gaps > 50 us in the last step (start offset ms, gap us, after -> before, last other-stream kernel ending inside the gap):")
prev = main[0]
hist = defaultdict(int)
for r in main[1:]:
    g = r[0] - prev[1]
    if g > 5e3:
        hist[min(int(g / 1e3).bit_length(), 12)] += 1
    if g > 50e3:
        inside = [o for o in others if prev[1] <= o[1] <= r[0] + 20e3]
        last = max(inside, key=lambda o: o[1]) if inside else None
        print("  %8.2f %7.0f  %s -> %s   | %s" % ((prev[1] - (t1 - ms * 1e6)) / 1e6, g / 1e3, family(prev[2]), family(r[2]),
                                                 ("%s (+%.0f us before the gap closed)" % (family(last[2]), (r[0] - last[1]) / 1e3)) if last else "-"))
    if r[1] > prev[1]:
        prev = r
print("gap histogram (us, power-of-two buckets):", {("<%d" % (1 << k)): v for k, v in sorted(hist.items())})

# optional: chronological listing of all streams for windows of the last step:  ... <db> <steps> <ms> a0:a1 [b0:b1 ...]  (ms)
for w in sys.argv[4:]:
    a, b = (float(v) for v in w.split(":"))
    base = t1 - ms * 1e6
    print("\nlast step, %.1f .. %.1f ms: (start, duration us, stream, kernel)" % (a, b))
    for s, e, n, _, q in win:
        if base + a * 1e6 <= s <= base + b * 1e6:
            print("  %8.3f %7.0f  s%s%s %s" % ((s - base) / 1e6, (e - s) / 1e3, q, "*" if q == conv_q else " ", n[:90]))
