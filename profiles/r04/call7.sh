set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/ab_env.sh ring 2 "PP_CONV_RING=0" "PP_CONV_RING=-1" | tee -a gpurun_out/ab_ring.txt
bash profiles/ab_env.sh strided_order 2 "PP_STRIDED_ORDER_RATIO=0" "PP_STRIDED_ORDER_RATIO=1.7" | tee -a gpurun_out/ab_strided_order.txt
cat gpurun_out/ab_ring.txt gpurun_out/ab_strided_order.txt
python -m pytest tests/test_model_gpu.py tests/test_hip_ops.py -m gpu -x -q -k "oracle or switches" 2>&1 | tail -3
