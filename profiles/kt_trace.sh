# kernel trace of the bench (3 steps) -> stream timeline: bash profiles/kt_trace.sh [tag]   (on the GPU box)
T=${1:-r03_kt}; O=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $O; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_kt
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks ${BENCH_ARGS:-} 2>/dev/null | tail -1 > $O/bench_kt.json
MS=$(python -c "import json; print(json.load(open('$O/bench_kt.json'))['ms_per_step'])")
echo ms_per_step $MS
python $GRAFT_REPO_ROOT/profiles/stream_timeline.py /tmp/p_kt/kt_results.db 3 $MS ${WIN:-90:115} > $O/timeline.txt
