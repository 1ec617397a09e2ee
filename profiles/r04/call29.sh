set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_scene_gpu.py -m gpu -x -q -k "proposals_unique or dedup or scene_labels or prepared or switches" 2>&1 | tail -6 | cut -c1-220
for rep in 1 2 3; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fused  ', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['config']['checks']['all'])"
PP_DEDUPE_FUSED=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('library', round(j['ms_per_step'],2), round(j['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/r04_ab_proposals_fused.txt
