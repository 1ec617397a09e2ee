set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py tests/test_scene_gpu.py tests/test_losses_gpu.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-200
python profiles/step_boundary_trace.py 6 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2 3; do python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['checks']['all'])"; done
