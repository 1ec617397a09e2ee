set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python profiles/proto/two_in_flight.py 2 6 0 full 2>&1 | grep -v "amdgpu.ids" | tail -3
timeout 900 python profiles/proto/two_in_flight.py 2 6 60 full 2>&1 | grep -v "amdgpu.ids" | tail -1
