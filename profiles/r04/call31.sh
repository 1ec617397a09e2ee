set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
bash profiles/ab_env.sh prep_prio 3 "PP_PREP_PRIORITY=0" "PP_PREP_PRIORITY=1" 2>&1 | tail -12
