#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of row gathers with KNOWN traffic for the row widths the convolutions gather (64 .. 384 bytes), random
# and slot-ordered (identity) -- profiles/gather_calib.py -> gpurun_out/<tag>_fetch_calibration.md + <tag>_fetch_factors.json
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O/cal_$TAG
cd /tmp && export TMPDIR=/tmp
for W in 64 128 192 256 384; do
for M in random ident; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_cal
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_cal -o c -- python $R/profiles/gather_calib.py $M $W > /dev/null 2>&1
  python $R/profiles/rocpd_summary.py --pmc /tmp/p_cal/c_results.db $O/cal_$TAG/${W}_${M}_$C.md > /dev/null 2>&1
done
done
done
python - <<PY
import json
def load(f):
    d = {}
    try:
        for l in open(f):
            p = [x.strip() for x in l.split("|")]
            if len(p) > 5 and p[3].isdigit():
                d.setdefault(p[1], []).append((p[2], int(p[3]), float(p[4])))
    except OSError:
        pass
    return d
out = ["# FETCH_SIZE / WRITE_SIZE calibration by row width (profiles/calibrate_fetch_widths.sh, profiles/gather_calib.py)", "",
       "k_gather_rows16 over a 2 GiB table of W-byte rows (n = 2 GiB / W rows, each fetched exactly once): known reads = n (W + 8) bytes",
       "(rows + int64 index), known writes = n W.  random = a random permutation (every row a separate request, far beyond the 256 MiB",
       "infinity cache), ident = the identity (the slot-ordered / streaming case).  rocprofv3 reports the counters in KiB.", "",
       "| row bytes | pattern | counter | dispatches | raw bytes per dispatch | known bytes | known / raw |", "|---|---|---|---|---|---|---|"]
factors = {}
for W in (64, 128, 192, 256, 384):
    n = (1 << 31) // W
    known = {"FETCH_SIZE": n * (W + 8), "WRITE_SIZE": n * W}
    for m in ("random", "ident"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for k, rows in load("$O/cal_$TAG/%d_%s_%s.md" % (W, m, c)).items():
                if "k_gather_rows16" in k:
                    for counter, nd, avg in rows:
                        raw = avg * 1024
                        out.append("| %d | %s | %s | %d | %.5g | %.5g | %.3f |" % (W, m, counter, nd, raw, known[c], known[c] / raw))
                        factors.setdefault(str(W), {})["%s_%s" % (m, c)] = known[c] / raw
open("$O/${TAG}_fetch_calibration.md", "w").write("\n".join(out) + "\n")
json.dump(factors, open("$O/${TAG}_fetch_factors.json", "w"), indent=1)
print("\n".join(out))
PY
