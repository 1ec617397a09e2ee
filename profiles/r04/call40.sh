set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-200 | tee gpurun_out/r04_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r04_final_tests.txt
bash profiles/collect.sh r04_final2 > gpurun_out/r04_final2_collect.log 2>&1
for i in 1 2 3; do python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard ms_per_step', j['ms_per_step'])"; done | tee gpurun_out/r04_shard_runs_final.txt
tail -c 400 gpurun_out/r04_final2/bench.json
