mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02b; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_kt
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r02b/bench_kt.json
cp /tmp/p_kt/kt_results.db $GRAFT_REPO_ROOT/gpurun_out/r02b/kt.db
python -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/r02b/bench_kt.json')); print('ms_per_step', d['ms_per_step'])"
