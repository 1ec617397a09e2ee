set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -k "compact or switches" 2>&1 | tail -12 | cut -c1-250
for rep in 1 2 3; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('compact', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['checks']['all'])"
PP_COMPACT_MAPS=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('dense  ', round(j['ms_per_step'],2), round(j['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/r04_ab_compact_maps.txt
