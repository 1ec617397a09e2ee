set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env $1 python bench.py --points 2000000 --grid 4 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2))"; }
for i in 1 2 3; do
run "PP_X=0"
run "PP_SAME_WINDOW=16384,32768"
run "PP_SEGMENT_DETERMINISTIC=0"
run "PP_DEDUPE_FUSED=0"
run "PP_INPUT_PREFETCH=0"
done | tee gpurun_out/r04_c2_ab.txt
