# round 6: k_spconv_x3f (full-line gathers through LDS) against k_spconv_x3 (PP_CONV_X3F=0): conv tests, then the bench step, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x3f
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or spconv or transposed_map" 2>&1 | tail -5 > gpurun_out/x3f/tests.txt
cat gpurun_out/x3f/tests.txt
for v in 1 0 1 0; do echo "== PP_CONV_X3F=$v"; PP_CONV_X3F=$v python bench.py --no-cpu-baseline --no-checks --steps 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('by_kernel_family',{})
print('ms_per_step %.2f single %.2f frac %.4f'%(d['ms_per_step'], d['config'].get('single_scene_ms',0), r['frac']), {k:round(v.get('ms_per_step',0),2) for k,v in f.items()})
"; done > gpurun_out/x3f/bench_ab.txt 2>&1
cat gpurun_out/x3f/bench_ab.txt
