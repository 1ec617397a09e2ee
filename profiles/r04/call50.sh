set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "prepared or run_to_run or training" 2>&1 | tail -12 | cut -c1-250
for i in 1 2 3; do for P in 1 0; do PP_TRAIN_INPUT_PREFETCH=$P python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('train prefetch=$P', [v['ms_per_step'] for v in j['modes'].values()])"; done; done
