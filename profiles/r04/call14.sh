set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -s -k "weight_gradient or training or wgrad or pair_lists or reproducible" > gpurun_out/r04_call14_tests.txt 2>&1
tail -15 gpurun_out/r04_call14_tests.txt | cut -c1-220
python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print({k: v['ms_per_step'] for k, v in j['modes'].items()})"
PP_WGRAD_DETERMINISTIC=0 python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('atomic:', {k: v['ms_per_step'] for k, v in j['modes'].items()})"
