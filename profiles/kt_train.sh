# kernel trace of the training step, first mode only -> stream timeline: bash profiles/kt_train.sh [tag]   (on the GPU box)
T=${1:-r03_kt_train}; O=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $O; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_ktt
PP_TRAIN_MODES=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/p_ktt -o kt -- python $GRAFT_REPO_ROOT/profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_kt.json
MS=$(python -c "import json; print(list(json.load(open('$O/train_kt.json'))['modes'].values())[0]['ms_per_step'])")
echo ms_per_step $MS
python $GRAFT_REPO_ROOT/profiles/stream_timeline.py /tmp/p_ktt/kt_results.db 4 $MS > $O/timeline_train.txt
python $GRAFT_REPO_ROOT/profiles/rocpd_summary.py /tmp/p_ktt/kt_results.db 70 > $O/kernel_stats_train.md 2>/dev/null || true
python - > $O/torch_kernels_train.txt <<PY
import sqlite3
con = sqlite3.connect("/tmp/p_ktt/kt_results.db")
rows = con.execute("select name, count(*), sum(end-start) from kernels group by name order by count(*) desc").fetchall()
steps = 7
print("launches per step and us per step by kernel (full names), all 7 steps of the traced run")
for n, c, t in rows:
    if c >= steps:
        print("%7.1f %8.1f  %s" % (c / steps, t / steps / 1e3, n[:420]))
PY
