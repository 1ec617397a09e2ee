#!/bin/bash
# SQ activity / wait counters per kernel for one bench step (which kernels sit in s_waitcnt, which issue instructions):
#   bash profiles/pmc_sq.sh [points] [grid]      -> gpurun_out/pmc_sq.md
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_sq
timeout 800 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS \
  -d /tmp/p_sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --points ${1:-3000000} --grid ${2:-5} > /dev/null 2>&1
mkdir -p $R/gpurun_out
python $R/profiles/rocpd_summary.py --pmc /tmp/p_sq/sq_results.db $R/gpurun_out/pmc_sq.md > /dev/null
