# round 6, final tree: whole GPU suite + smoke + the bench line + the self-launched gloo world-2 run (both ranks on the one GPU)
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/r06f
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/r06f/final_tests.txt
cat gpurun_out/r06f/final_tests.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r06f/bench.json
PP_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 > gpurun_out/r06f/gloo_world2.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06f/bench.json")); r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "traffic", r.get("traffic_over_algorithmic"), r.get("traffic_bounds"))
g=json.load(open("gpurun_out/r06f/gloo_world2.json")); print("gloo", g["n_gpus"], g["ms_per_step"], g["config"]["multi_gpu"] is not None)
PY
