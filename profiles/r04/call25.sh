set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_scene_gpu.py -m gpu -x -q -k "prepared" 2>&1 | tail -15 | cut -c1-220
for rep in 1 2 3; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('prefetch', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['checks']['all'], j['config']['host']['step_ms'])"
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-checks --no-input-prefetch 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('plain   ', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['host']['step_ms'])"
done 2>&1 | tee gpurun_out/r04_ab_input_prefetch.txt
