#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of launches with known traffic (profiles/gather_calib.py) -> gpurun_out/<tag>_fetch_calibration.md
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in random ident; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_cal
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_cal -o c -- python $R/profiles/gather_calib.py $M > /dev/null 2>&1
  python $R/profiles/rocpd_summary.py --pmc /tmp/p_cal/c_results.db $O/${TAG}_cal_${M}_$C.md > /dev/null
done
done
python - <<PY
import re
def load(f):
    d = {}
    for l in open(f):
        p = [x.strip() for x in l.split("|")]
        if len(p) > 5 and p[3].isdigit():
            d.setdefault(p[1], []).append((p[2], int(p[3]), float(p[4])))
    return d
out = ["# FETCH_SIZE / WRITE_SIZE calibration (profiles/calibrate_fetch.sh, profiles/gather_calib.py)", "",
       "Known traffic per launch: triad reads 2 x 1.074 GB and writes 1.074 GB; gather64 (random permutation of 32 M 64-byte rows of a 2 GiB",
       "table) reads 2.147 GB of rows + 0.268 GB of index and writes 2.147 GB; gather64s is the same kernel with the identity permutation.",
       "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.", "",
       "| kernel | counter | dispatches | avg per dispatch (raw, KiB) | raw bytes | known bytes | known / raw |", "|---|---|---|---|---|---|---|"]
known = {"k_triad": {"FETCH_SIZE": 2 * 4 * (1 << 28), "WRITE_SIZE": 4 * (1 << 28)},
         "k_gather_rows16": {"FETCH_SIZE": (1 << 25) * (64 + 8), "WRITE_SIZE": (1 << 25) * 64}}
for m in ("random", "ident"):
  for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, rows in load("$O/${TAG}_cal_%s_%s.md" % (m, c)).items():
        for kk, kb in known.items():
            if kk in k:
                for counter, n, avg in rows:
                    raw = avg * 1024
                    out.append("| %s (%s) | %s | %d | %.6g | %.4g | %.4g | %.3f |" % (k[:40], m if "gather" in k else "-", counter, n, avg, raw, kb[c], kb[c] / raw))
open("$O/${TAG}_fetch_calibration.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
