# round 6 evidence set on one GPU box: bash profiles/final_evidence_r06.sh   -> gpurun_out/r06/*
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
ulimit -c 0
bash profiles/collect.sh r06 > $O/collect.log 2>&1
WIN=85:112 bash profiles/kt_trace.sh r06_kt > $O/kt.log 2>&1
SHARD_TAG=r06_shard bash profiles/kt_shard.sh > $O/shard.log 2>&1
python bench.py --points 1250000 --grid 3 --steps 20 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 > $O/shard_bench.json
python profiles/config_microbench.py --out $O > $O/configs.log 2>&1
python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_fp32.json
PP_CONV_DTYPE=bf16 python profiles/train_microbench.py 2>/dev/null | tail -1 > $O/train_bf16.json
ls $O
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench.json")); r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "frac_pipe", r.get("frac_pipe"), "traffic/alg", r.get("traffic_over_algorithmic"))
print({k:(round(v["ms_per_step"],2), round(v["frac_pipe"],3), round(v.get("traffic_over_algorithmic",0),2)) for k,v in r["by_kernel_family"].items()})
print(d["cpu_baseline"]); print(d["config"].get("checks"))
PY
