set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py tests/test_losses_gpu.py tests/test_forest_gpu.py tests/test_hip_ops.py -m gpu -x -q -k "training or train or loss or grad or autocast or scatter or segment" 2>&1 | tail -3 | cut -c1-220
for i in 1 2 3; do python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('train', [v['ms_per_step'] for v in j['modes'].values()])"; done
