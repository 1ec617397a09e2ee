#!/bin/bash
# One bench.py run per environment string inside ONE gpurun call (same box):
#   bash profiles/sweep_env.sh <tag> "<ENV_1>" "<ENV_2>" ...     -> gpurun_out/sweep_<tag>.txt
# one line per run: ms_per_step, roofline.frac, convolution ms per step (all / x3 family / fwd3 family)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sweep_$TAG.txt
: > $O
for E in "$@"; do
  env $E python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-checks ${BENCH_ARGS:-} 2>$R/gpurun_out/sweep_${TAG}_err.log | tail -1 | \
    python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; f=r.get('by_kernel_family') or {}
print('$E', '| ms_per_step %.2f' % j['ms_per_step'], 'single %.2f' % (j['config'].get('single_scene_ms') or 0), 'frac %.4f' % r['frac'],
      'conv_ms %.2f' % (r['avg_launch_us'] * r['launches_per_step'] / 1e3),
      ' '.join('%s %.2f' % (k, v['ms_per_step']) for k, v in sorted(f.items())))" >> $O
done
cat $O
