set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -c 2500 gpurun_out/r04_bench_a.json
python -m pytest tests -m gpu -x -q > gpurun_out/r04_call4_tests.txt 2>&1
tail -5 gpurun_out/r04_call4_tests.txt
