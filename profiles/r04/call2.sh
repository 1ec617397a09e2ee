set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_edge_cases_gpu.py tests/test_eval_gpu.py tests/test_scene_gpu.py tests/test_formats_gpu.py tests/test_model_gpu.py -m gpu -x -q -s -k "no_grad or storage_swap or frozen_second or treeins or tracker_batch or scene_labels or eval_ply or deduplication or without_score" > gpurun_out/r04_call2_tests.txt 2>&1
tail -30 gpurun_out/r04_call2_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_call2_smoke.txt 2>&1; tail -3 gpurun_out/r04_call2_smoke.txt
