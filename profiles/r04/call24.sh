set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "hdbscan or hd_" 2>&1 | tail -3 | cut -c1-200
python profiles/host_step_profile.py 10000000 8 6 > gpurun_out/r04_host_step_profile_n1.txt 2>&1
head -50 gpurun_out/r04_host_step_profile_n1.txt | cut -c1-200
