# round 6: state of the step with k_spconv_x3f on every whole-group layer incl. one column tile: layer table, kernel stats, timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2b
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or spconv or transposed_map or kernel_family" 2>&1 | tail -3 > gpurun_out/s2b/tests.txt
cat gpurun_out/s2b/tests.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-checks --stage-timing --layer-table gpurun_out/s2b/layer_table.md 2>/dev/null | tail -1 > gpurun_out/s2b/bench_stages.json
python -c "
import json
d=json.load(open('gpurun_out/s2b/bench_stages.json')); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'frac', r['frac'], {k:round(v.get('ms_per_step',0),2) for k,v in r.get('by_kernel_family',{}).items()})
print(d['config'].get('stage_ms'))
"
bash profiles/kt_trace.sh s2b_kt
python profiles/rocpd_summary.py /tmp/p_kt/kt_results.db gpurun_out/s2b/kernel_stats.md > /dev/null
head -60 gpurun_out/s2b_kt/timeline.txt
