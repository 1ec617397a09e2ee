cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2j
P='
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("ms_per_step %.2f frac %.4f single %s"%(d["ms_per_step"], r["frac"], d["config"]["single_scene_ms"]))
'
for v in "" SCPQ CSPQ SPQC PQSC "" SCPQ CSPQ; do echo "== PP_STREAM_ORDER=[$v]"; PP_STREAM_ORDER=$v timeout 600 python bench.py --no-cpu-baseline --no-checks --steps 8 2>/dev/null | tail -1 | python -c "$P"; done > gpurun_out/s2j/stream_order.txt 2>&1
cat gpurun_out/s2j/stream_order.txt
