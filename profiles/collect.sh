#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>
# writes gpurun_out/<tag>/{bench.json, layer_table.md, kernel_stats.md, pmc_fetch.md, pmc_write.md, traffic.json};
# copy them to profiles/.  --pmc runs are separate passes with --kernel-trace only (gpurun refuses pmc + sys/runtime trace).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w
# PMC passes first: bench.py's roofline.traffic reads profiles/traffic.json, which must come from this tree
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-checks > /dev/null 2>&1
python $R/profiles/rocpd_summary.py --pmc /tmp/p_f/f_results.db $O/pmc_fetch.md > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-checks > /dev/null 2>&1
python $R/profiles/rocpd_summary.py --pmc /tmp/p_w/w_results.db $O/pmc_write.md > /dev/null
python $R/profiles/rocpd_summary.py --traffic /tmp/p_f/f_results.db /tmp/p_w/w_results.db $O/traffic.json
cp $O/traffic.json $R/profiles/traffic.json
timeout 900 python $R/bench.py 2>/dev/null | tail -1 > $O/bench.json
timeout 900 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --stage-timing --layer-table $O/layer_table.md 2>/dev/null | tail -1 > $O/bench_stages.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks > /dev/null 2>&1
python $R/profiles/rocpd_summary.py /tmp/p_kt/kt_results.db $O/kernel_stats.md > /dev/null
ls -la $O
