set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/final_evidence.sh > gpurun_out/r04_final_evidence.log 2>&1
SHARD_TAG=r04_shard bash profiles/kt_shard.sh > gpurun_out/r04_shard_kt.log 2>&1
python profiles/count_syncs.py > gpurun_out/r04_shard_host_syncs.txt 2>&1
python profiles/rocpd_summary.py /tmp/p_kt/kt_results.db gpurun_out/r04_shard/kernel_stats_shard.md > /dev/null 2>&1
for i in 1 2 3; do python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'])"; done > gpurun_out/r04_shard_runs.txt
bash profiles/clock_probe.sh r04 64 1:16:16,2:32:32,4:48:48,8:64:64 > /dev/null 2>&1
ls gpurun_out/r04_final gpurun_out/r04_shard; cat gpurun_out/r04_shard_runs.txt; tail -c 600 gpurun_out/r04_final/bench.json
