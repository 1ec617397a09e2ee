set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --stage-timing --layer-table gpurun_out/r04_layer_table_a.md 2>/dev/null | tail -1 > gpurun_out/r04_bench_stages_a.json
python -c "
import json
j=json.loads(open('gpurun_out/r04_bench_stages_a.json').read()); print(j['ms_per_step'], j['config'].get('stage_ms'))"
cat gpurun_out/r04_layer_table_a.md
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['frac'])"
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'], j['config']['host']['step_ms'])"
timeout 600 python profiles/host_profile.py 1250000 > gpurun_out/r04_shard_host_profile.txt 2>&1; grep -n "was called by" -A14 gpurun_out/r04_shard_host_profile.txt | cut -c1-200 | head -80
