set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python profiles/host_step_profile.py 1250000 3 20 > gpurun_out/r04_shard_host_step_profile.txt 2>&1
cut -c1-200 gpurun_out/r04_shard_host_step_profile.txt | head -120
