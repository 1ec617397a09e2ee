set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['frac'])"
for i in 1 2; do
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'], j['config']['host']['step_ms'])"
done
python bench.py --points 2000000 --grid 4 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('C2 ms_per_step', j['ms_per_step'], j['config']['workload'][:80])"
python -m pytest tests -m gpu -x -q > gpurun_out/r04_call12_tests.txt 2>&1
tail -5 gpurun_out/r04_call12_tests.txt
