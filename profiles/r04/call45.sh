set -u
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for L in 1 0; do PP_LINEAR_ROWS=$L python profiles/train_microbench.py 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('linear_rows=$L', [v['ms_per_step'] for v in j['modes'].values()])"; done; done
