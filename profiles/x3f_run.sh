cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2k
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "^E |Error|FAILED|passed|failed" | head -30 > gpurun_out/s2k/first_fail.txt
cat gpurun_out/s2k/first_fail.txt
