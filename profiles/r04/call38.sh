set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -k "compact or switches" 2>&1 | tail -6 | cut -c1-250
