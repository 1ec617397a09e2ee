set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_scene_gpu.py tests/test_forest_gpu.py tests/test_variants_gpu.py tests/test_nms_gpu.py -m gpu -x -q > gpurun_out/r04_call16_tests.txt 2>&1
tail -5 gpurun_out/r04_call16_tests.txt | cut -c1-200
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['frac'], j['config']['checks']['all'])"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --stage-timing 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['config'].get('stage_ms'))"
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'])"
