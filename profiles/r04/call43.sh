set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
for P in default high; do
PP_SIDE_PRIORITY=$P python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard $P', round(j['ms_per_step'],2))"
PP_SIDE_PRIORITY=$P python bench.py --points 2000000 --grid 4 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('c2 $P', round(j['ms_per_step'],2))"
done
done | tee gpurun_out/r04_ab_side_priority_small.txt
bash profiles/collect.sh r04_final3 > gpurun_out/r04_final3_collect.log 2>&1
python -m pytest tests/test_model_gpu.py tests/test_scene_gpu.py -m gpu -x -q 2>&1 | tail -2
tail -c 300 gpurun_out/r04_final3/bench.json
