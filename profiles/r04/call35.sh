set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "run_to_run" 2>&1 | grep -v amdgpu | tail -14 | cut -c1-600
