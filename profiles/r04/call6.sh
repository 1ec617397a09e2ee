set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_fullsize_gpu.py -m gpu -x -q -k "conv or shortcut or transposed or t8 or 8_wide" > gpurun_out/r04_call6_tests.txt 2>&1
tail -4 gpurun_out/r04_call6_tests.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
python -c "
import json
j=json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
print('ring default:', j['ms_per_step'], j['roofline']['frac'], j['config']['checks']['all'])"
PP_CONV_RING=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks > gpurun_out/r04_bench_b0.json 2>/dev/null
python -c "
import json
j=json.loads(open('gpurun_out/r04_bench_b0.json').read().strip().splitlines()[-1])
print('ring off:', j['ms_per_step'], j['roofline']['frac'])"
bash profiles/kt_trace.sh r04_kt > /dev/null 2>&1; head -60 gpurun_out/r04_kt/timeline.txt
