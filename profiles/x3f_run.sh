# round 6: backbone ahead with / without the scorer's convolutions waiting for it: share, configs[1], full scene
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2q
P='
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("ms_per_step %.2f frac %.4f"%(d["ms_per_step"], r["frac"]))
'
for v in 0 1 0 1; do echo "== share PP_AHEAD_SCORER_WAIT=$v"; PP_AHEAD_SCORER_WAIT=$v timeout 600 python bench.py --points 1250000 --grid 3 --steps 30 --warmup 4 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done > gpurun_out/s2q/scorer_wait.txt 2>&1
for v in 0 1 0 1; do echo "== C2 PP_AHEAD_SCORER_WAIT=$v"; PP_AHEAD_SCORER_WAIT=$v timeout 600 python bench.py --points 2000000 --grid 4 --steps 20 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done >> gpurun_out/s2q/scorer_wait.txt 2>&1
for v in 0 1; do echo "== full (--backbone-ahead on) PP_AHEAD_SCORER_WAIT=$v"; PP_AHEAD_SCORER_WAIT=$v timeout 600 python bench.py --steps 8 --backbone-ahead on --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done >> gpurun_out/s2q/scorer_wait.txt 2>&1
cat gpurun_out/s2q/scorer_wait.txt
