set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "map_order" 2>&1 | tail -2 | cut -c1-200
for rep in 1 2 3; do
for W in 8192,8192 8192,32768 16384,32768 8192,16384; do
PP_SAME_WINDOW=$W python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('SW=$W', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['roofline']['achieved'])"
done
done 2>&1 | tee gpurun_out/r04_same_window_sweep2.txt
PP_MAP_WINDOW=4096 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-checks 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -8 | cut -c1-300
