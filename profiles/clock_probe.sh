#!/bin/bash
# Effective shader clock inside the convolution launches: GRBM_GUI_ACTIVE (cycles the GPU was busy, shader clock) per
# dispatch / the dispatch's duration (MI355X_MICROARCH.md "DVFS give-back"), next to the MFMA pipe's busy cycles.
#   bash profiles/clock_probe.sh <tag> <n_tiles> <shapes as in conv_one.py>     -> gpurun_out/clock_<tag>.md
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clock_$TAG.md
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk_$TAG
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d /tmp/clk_$TAG -o p -- python $R/profiles/conv_one.py "$@" 3 > /tmp/clk_$TAG.log 2>&1
python - "$O" /tmp/clk_$TAG/p_results.db <<'PY'
import sqlite3, sys
out, db = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
lines = ["# effective clock per convolution dispatch (GRBM_GUI_ACTIVE / duration)", ""]
try:
    rows = con.execute("select c.dispatch_id, c.kernel_name, c.counter_name, c.value, k.duration, k.grid_x from counters_collection c "
                       "join kernels k on k.dispatch_id = c.dispatch_id where c.kernel_name like '%spconv%' order by c.dispatch_id").fetchall()
    by = {}
    for d, name, cn, v, dur, gx in rows:
        e = by.setdefault(d, {"name": name, "dur": dur, "grid": gx})
        e[cn] = e.get(cn, 0) + v
    lines += ["| dispatch | grid_x | us | GRBM_GUI_ACTIVE | GHz | SQ_BUSY_CYCLES | MFMA busy / (GUI_ACTIVE x 1024 SIMDs) | MFMA insts |", "|---|---|---|---|---|---|---|---|"]
    for d, e in sorted(by.items()):
        gui = e.get("GRBM_GUI_ACTIVE", 0)
        lines.append("| %d | %d | %.1f | %.4g | %.3f | %.4g | %.3f | %.4g |" % (d, e["grid"], e["dur"] / 1e3, gui, gui / max(e["dur"], 1), e.get("SQ_BUSY_CYCLES", 0),
                                                                   e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui * 1024.0, 1), e.get("SQ_INSTS_MFMA", 0)))
except Exception as ex:  # schema differs: dump what is there
    lines.append("query failed: %r; tables: %s" % (ex, tabs))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
tail -20 /tmp/clk_$TAG.log >> $O
