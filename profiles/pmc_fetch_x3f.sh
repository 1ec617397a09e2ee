#!/bin/bash
# round 6: how many bytes does one 64 -> 64 layer fetch from memory on k_spconv_x3f (full 128-byte line requests) and on k_spconv_x3
# (64-byte pieces)?  FETCH_SIZE (KiB, = 64 B x read requests) beside the request counters themselves, separate --pmc passes.
#   bash profiles/pmc_fetch_x3f.sh   -> gpurun_out/pmc_fetch_x3f/summary.md
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_fetch_x3f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/summary.md
for F in 1 0; do
  i=0
  for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rm -rf /tmp/pf_$F$i
    PP_CONV_X3F=$F timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pf_$F$i -o p -- python $R/profiles/conv_one.py 64 8:64:64,4:64:64 2 > /dev/null 2>&1
    python $R/profiles/rocpd_summary.py --pmc /tmp/pf_$F$i/p_results.db $O/x3f${F}_pass$i.md > /dev/null 2>&1 || echo "pass $F $i failed ($C)" >> $O/summary.md
  done
  echo "== PP_CONV_X3F=$F" >> $O/summary.md
  grep -h spconv $O/x3f${F}_pass*.md >> $O/summary.md
done
cat $O/summary.md | cut -c1-200
