set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for W in 8192,8192 8192,16384 8192,32768 16384,16384 16384,32768; do
rm -rf /tmp/p_f
PP_SAME_WINDOW=$W timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-checks > /dev/null 2>&1
python $R/profiles/rocpd_summary.py --pmc /tmp/p_f/f_results.db /tmp/pmc_fetch_$W.md > /dev/null
grep "k_spconv_fwd" /tmp/pmc_fetch_$W.md | awk -F'|' -v w=$W '{calls+=$4; kib+=$6} END {printf "SW=%s conv launches %d  FETCH_SIZE raw %.1f MB per launch\n", w, calls, kib/calls/1024}'
done 2>&1 | tee $R/gpurun_out/r04_window_traffic.txt
