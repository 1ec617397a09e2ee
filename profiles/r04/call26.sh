set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WIN=-3:12 bash profiles/kt_trace.sh r04_kt2 > /dev/null 2>&1
sed -n 1,45p gpurun_out/r04_kt2/timeline.txt | cut -c1-200
sed -n '/main-stream gaps/,/gap histogram/p' gpurun_out/r04_kt2/timeline.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard prefetch ms_per_step', j['ms_per_step'])"
python bench.py --points 1250000 --grid 3 --steps 10 --warmup 3 --no-cpu-baseline --no-checks --no-input-prefetch 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('shard plain    ms_per_step', j['ms_per_step'])"
done | tee gpurun_out/r04_shard_runs2.txt
