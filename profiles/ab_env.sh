#!/bin/bash
# Alternating A/B runs of bench.py under two environments inside ONE gpurun call (boxes differ by +-1 %, runs on one box by
# +-0.3 %):   bash profiles/ab_env.sh <tag> <pairs> "<ENV_A>" "<ENV_B>"     e.g.  ... r04_ring 3 "PP_CONV_RING=0" "PP_CONV_RING=-1"
# -> gpurun_out/ab_<tag>.txt: one line per run (ms_per_step, roofline.frac) and the means
set -u
TAG=$1; PAIRS=$2; A=$3; B=$4
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_$TAG.txt
: > $O
for i in $(seq 1 $PAIRS); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    env $E python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$side', '$E', 'ms_per_step %.2f' % j['ms_per_step'], 'frac %.4f' % j['roofline']['frac'], 'conv_ms %.2f' % (j['roofline']['avg_launch_us'] * j['roofline']['launches_per_step'] / 1e3))" >> $O
  done
done
python - $O <<'PY'
import sys
rows=[l.split() for l in open(sys.argv[1])]
for side in "AB":
    ms=[float(r[r.index('ms_per_step')+1]) for r in rows if r[0]==side]
    cv=[float(r[r.index('conv_ms')+1]) for r in rows if r[0]==side]
    print(side, "mean ms_per_step %.2f (min %.2f max %.2f)  conv %.2f" % (sum(ms)/len(ms), min(ms), max(ms), sum(cv)/len(cv)))
PY
