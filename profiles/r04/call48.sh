set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "meanshift or mean_shift or cluster or model or scene or forest or embed" 2>&1 | tail -2 | cut -c1-200
python profiles/infer_sync_trace.py 2>&1 | grep -v amdgpu.ids | grep "meanshift\|synchronising"
for i in 1 2 3; do
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard', round(j['ms_per_step'],2))"
done
