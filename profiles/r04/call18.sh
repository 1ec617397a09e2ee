set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "conv or input or oracle or shapes or fullsize or one_hot" 2>&1 | tail -4 | cut -c1-200
for i in 1 2; do
PP_HIP_LIB=$PWD/profiles/abl/libpanoptic_base.so python profiles/conv_one.py 64 1:4:16 5 2>&1 | grep "ts=" | sed "s/^/base: /"
python profiles/conv_one.py 64 1:4:16 5 2>&1 | grep "ts=" | sed "s/^/new:  /"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['frac'], j['config']['checks']['all'])"
