set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | cut -c1-250 | tee gpurun_out/r04_call34_tests.txt
python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print({k:v['ms_per_step'] for k,v in j['modes'].items()})"
PP_SEGMENT_DETERMINISTIC=0 python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('atomic segment sums', {k:v['ms_per_step'] for k,v in j['modes'].items()})"
