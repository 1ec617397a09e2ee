mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${SHARD_TAG:-r03_shard}; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_kt
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --points 1250000 --grid 3 --steps 5 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${SHARD_TAG:-r03_shard}/bench_kt.json
MS=$(python -c "import json; print(json.load(open('$GRAFT_REPO_ROOT/gpurun_out/${SHARD_TAG:-r03_shard}/bench_kt.json'))['ms_per_step'])")
echo ms_per_step $MS
python $GRAFT_REPO_ROOT/profiles/stream_timeline.py /tmp/p_kt/kt_results.db 5 $MS -2:40 > $GRAFT_REPO_ROOT/gpurun_out/${SHARD_TAG:-r03_shard}/timeline_shard.txt
