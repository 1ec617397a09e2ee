set -u
cd $GRAFT_REPO_ROOT
PMC_PASSES=4 bash profiles/pmc_conv.sh r04_c16 64 1:16:16 > /dev/null 2>&1
PMC_PASSES=4 bash profiles/pmc_conv.sh r04_c64 64 8:64:64 > /dev/null 2>&1
PMC_PASSES=4 bash profiles/pmc_conv.sh r04_c48 64 4:48:48 > /dev/null 2>&1
for t in c16 c64 c48; do echo "== $t"; head -3 gpurun_out/pmc_r04_$t/summary.md; grep -c spconv gpurun_out/pmc_r04_$t/summary.md; done
