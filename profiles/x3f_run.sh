# round 6: block-level same-map lookup (pp_kernel_map_bi_same): parity, stand-alone stage times, bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2c
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "block_level or block_index or level_chain or kernel_maps" 2>&1 | tail -3 > gpurun_out/s2c/tests.txt
cat gpurun_out/s2c/tests.txt
python profiles/map_build_one.py 64 5 2>/dev/null | grep -v amdgpu.ids > gpurun_out/s2c/map_build_one.txt
cat gpurun_out/s2c/map_build_one.txt
for v in 1 0 1 0; do echo "== PP_MAP_BLOCKS=$v"; PP_MAP_BLOCKS=$v python bench.py --no-cpu-baseline --no-checks --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('by_kernel_family',{})
print('ms_per_step %.2f single %.2f frac %.4f'%(d['ms_per_step'], d['config'].get('single_scene_ms',0), r['frac']), {k:round(v.get('ms_per_step',0),2) for k,v in f.items()})
"; done > gpurun_out/s2c/bench_ab.txt 2>&1
cat gpurun_out/s2c/bench_ab.txt
