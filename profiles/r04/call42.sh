set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PP_DIST_BACKEND=gloo PP_BENCH_SINGLE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --points 2000000 --grid 4 --no-cpu-baseline 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -3 | cut -c1-1500 | tee gpurun_out/r04_gloo_world2_dryrun.json
