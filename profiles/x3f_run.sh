cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2f
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/s2f/tests.txt
cat gpurun_out/s2f/tests.txt
