set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "training_gradients or deduplication or oracle" 2>&1 | tail -3 | cut -c1-200
bash profiles/ab_env.sh dedupe 2 "PP_DEDUPE_HASH=0" "PP_DEDUPE_HASH=1"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --stage-timing 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['config'].get('stage_ms'))"
