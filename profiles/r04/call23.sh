set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for W in 8192 16384 32768; do
PP_MAP_WINDOW=$W python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('MW=$W', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['roofline']['achieved'])"
done
done 2>&1 | tee gpurun_out/r04_cross_window_sweep.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-200 | tee gpurun_out/r04_call23_tests.txt
