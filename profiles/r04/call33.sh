set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_losses_gpu.py -m gpu -x -q -k "segment or run_to_run or loss or scatter" 2>&1 | tail -8 | cut -c1-250
for rep in 1 2 3; do
for W in 8192,8192 8192,16384 16384,32768; do
PP_SAME_WINDOW=$W python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('SW=$W', round(j['ms_per_step'],2), round(j['roofline']['frac'],4))"
done
done 2>&1 | tee gpurun_out/r04_same_window_sweep3.txt
