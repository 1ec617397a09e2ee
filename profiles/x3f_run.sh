# round 6: HIP streams share hardware queues (default 4 per process): GPU_MAX_HW_QUEUES=8 against the default, share and full scene
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2h
P='
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; f=r.get("by_kernel_family",{})
print("ms_per_step %.2f single %.2f frac %.4f"%(d["ms_per_step"], d["config"].get("single_scene_ms",0), r["frac"]), {k:round(v.get("ms_per_step",0),2) for k,v in f.items()})
'
for v in 8 4 8 4; do echo "== share GPU_MAX_HW_QUEUES=$v"; GPU_MAX_HW_QUEUES=$v python bench.py --points 1250000 --grid 3 --steps 20 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done > gpurun_out/s2h/hwq.txt 2>&1
for v in 8 4 8 4; do echo "== full GPU_MAX_HW_QUEUES=$v"; GPU_MAX_HW_QUEUES=$v python bench.py --steps 8 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done >> gpurun_out/s2h/hwq.txt 2>&1
cat gpurun_out/s2h/hwq.txt
