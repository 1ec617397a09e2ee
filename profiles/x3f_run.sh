cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/r06f
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/r06f/final_tests.txt
cat gpurun_out/r06f/final_tests.txt
