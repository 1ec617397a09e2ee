set -u
cd $GRAFT_REPO_ROOT
python profiles/infer_sync_trace.py 2>&1 | grep -v amdgpu.ids | grep "MainThread\|synchronising"
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "region or grow or cluster or model or full" 2>&1 | tail -2 | cut -c1-200
for i in 1 2 3; do
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard', round(j['ms_per_step'],2))"
done
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['checks']['all'])"
