#!/bin/bash
# PMC passes for one convolution shape: bash profiles/pmc_conv.sh <tag> <n_tiles> <ts> <cin> <cout> <dense|rb>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/profiles/conv_one.py "$@" 2>/dev/null | tail -1 > $O/time.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm_$i -o p -- python $R/profiles/conv_one.py "$@" 2 > /dev/null 2>&1
  python $R/profiles/rocpd_summary.py --pmc /tmp/pm_$i/p_results.db $O/pass$i.md > /dev/null 2>&1 || echo "pass $i failed ($C)" >> $O/time.txt
done
cat $O/time.txt
grep -h spconv $O/pass*.md
