"""Where does the GPU idle?  Gaps between consecutive dispatches of a rocprofv3 kernel trace, grouped by the kernel that
FOLLOWS the gap (= first launch after a host synchronisation / Python-side work).
usage: python profiles/gap_analysis.py <results.db> <steps> <ms_per_step>"""
import sqlite3
import sys
from collections import defaultdict

db, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
con = sqlite3.connect(db)
rows = con.execute("select start, end, name from kernels order by start").fetchall()
tri = [i for i, r in enumerate(rows) if "k_triad" in r[2]]
rows = rows[: tri[0]] if tri else rows
t1 = rows[-1][1]
t0 = t1 - steps * ms * 1e6
win = [r for r in rows if r[0] >= t0]
gaps = defaultdict(lambda: [0, 0.0])
prev_end = win[0][1]
prev_name = win[0][2]
tot = 0.0
for s, e, name in win[1:]:
    g = s - prev_end
    if g > 20e3:  # > 20 us
        key = (prev_name[:50], name[:50])
        gaps[key][0] += 1
        gaps[key][1] += g
        tot += g
    prev_end = max(prev_end, e)
    prev_name = name
print("idle in gaps > 20 us: %.2f ms/step over %d dispatches/step" % (tot / steps / 1e6, len(win) / steps))
for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6.2f ms/step  %5.1f x/step  after [%s] before [%s]" % (g / steps / 1e6, c / steps, a, b))
