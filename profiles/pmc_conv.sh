#!/bin/bash
# PMC passes for convolution shapes: bash profiles/pmc_conv.sh <tag> <n_tiles> <shape>   (shape as in profiles/conv_one.py)
# (round 5: at most 4 TCC / TCP / TA counters per pass -- the 8-counter passes of round 4 failed: a block has 4 slots)
# -> gpurun_out/pmc_<tag>/{time.txt,passN.md,summary.md}.  Counter passes run with --kernel-trace only.
set -u
TAG=$1; shift
PROG=${PMC_PROG:-conv_one.py}   # PMC_PROG=wgrad_one.py: the weight-gradient shapes (arguments: <shape list>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/profiles/$PROG "$@" 5 2>/dev/null | tail -${PMC_TAIL:-2} > $O/time.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  [ $i -gt ${PMC_PASSES:-8} ] && break   # PMC_PASSES=2: the two SQ passes only (the TCC / TCP passes take minutes each)
  rm -rf /tmp/pm_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm_$i -o p -- python $R/profiles/$PROG "$@" 2 > /dev/null 2>&1
  python $R/profiles/rocpd_summary.py --pmc /tmp/pm_$i/p_results.db $O/pass$i.md > /dev/null 2>&1 || echo "pass $i failed ($C)" >> $O/time.txt
done
( cat $O/time.txt; echo; echo "| kernel | counter | dispatches | avg per dispatch | sum |"; echo "|---|---|---|---|---|"; grep -h spconv $O/pass*.md ) > $O/summary.md
cat $O/summary.md
