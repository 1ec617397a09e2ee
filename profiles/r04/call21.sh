set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "map_order" 2>&1 | tail -4 | cut -c1-200
for rep in 1 2; do
for W in 8192 16384 32768; do
PP_SAME_WINDOW=$W python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('SW=$W', round(j['ms_per_step'],2), round(j['roofline']['frac'],4), j['roofline']['achieved'], j['config']['checks']['all'])"
done
done 2>&1 | tee gpurun_out/r04_same_window_sweep.txt
PP_SAME_WINDOW=32768 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-checks --layer-table gpurun_out/r04_layer_table_sw32k.md >/dev/null 2>&1
