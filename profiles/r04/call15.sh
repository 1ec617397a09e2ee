set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -s -k "variants_match or training_gradients or weight_gradient_without" > gpurun_out/r04_call15_tests.txt 2>&1
tail -8 gpurun_out/r04_call15_tests.txt | cut -c1-220; grep -n "bit-identical run to run" gpurun_out/r04_call15_tests.txt
