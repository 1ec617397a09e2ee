# round 6: refresh of the records that --backbone-ahead auto changes: configs[1] / [2], one rank's share (line + traced timeline)
cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/r06b; mkdir -p $O
python profiles/config_microbench.py --out $O > $O/configs.log 2>&1
python bench.py --points 1250000 --grid 3 --steps 30 --warmup 4 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 > $O/shard_bench.json
SHARD_TAG=r06_shard_ahead bash profiles/kt_shard.sh > $O/shard.log 2>&1
python - <<'PY'
import json
for f in ("c2","c3","shard_bench"):
    d=json.load(open("gpurun_out/r06b/%s.json"%f)); print(f, d.get("ms_per_step"), d.get("value"), d.get("config",{}).get("backbone_ahead"))
PY
head -8 gpurun_out/r06_shard_ahead/timeline_shard.txt; grep -n "largest idle gaps" -A6 gpurun_out/r06_shard_ahead/timeline_shard.txt | head -12
