set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WIN=86:101.5 bash profiles/kt_trace.sh r04_kt3 > /dev/null 2>&1
grep -n "last step, 86" -A400 gpurun_out/r04_kt3/timeline.txt | grep " s0\* " | awk '{printf "%s %s %s\n", $2, $3, substr($0, index($0,$5), 70)}' | head -250
