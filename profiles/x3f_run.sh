# round 6: the backbone ahead enqueued by a host thread of its own (PP_AHEAD_THREAD=1) against the calling thread: parity, share, configs[1], full
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2r
timeout 900 python -m pytest tests/test_scene_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/s2r/tests.txt
cat gpurun_out/s2r/tests.txt
P='
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("ms_per_step %.2f frac %.4f"%(d["ms_per_step"], r["frac"]))
'
for v in 1 0 1 0; do echo "== share PP_AHEAD_THREAD=$v"; PP_AHEAD_THREAD=$v timeout 600 python bench.py --points 1250000 --grid 3 --steps 30 --warmup 4 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done > gpurun_out/s2r/ahead_thread.txt 2>&1
for v in 1 0 1 0; do echo "== C2 PP_AHEAD_THREAD=$v"; PP_AHEAD_THREAD=$v timeout 600 python bench.py --points 2000000 --grid 4 --steps 20 --warmup 3 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"; done >> gpurun_out/s2r/ahead_thread.txt 2>&1
cat gpurun_out/s2r/ahead_thread.txt
