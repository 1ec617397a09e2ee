set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_losses_gpu.py tests/test_forest_gpu.py -m gpu -x -q -k "linear or training or train or loss or grad or autocast" 2>&1 | tail -4 | cut -c1-220
for i in 1 2; do python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print({k:v['ms_per_step'] for k,v in j['modes'].items()})"; done
