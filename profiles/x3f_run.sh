cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2n
PP_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --points 2000000 --grid 4 --steps 5 --warmup 2 --no-cpu-baseline --no-checks 2>gpurun_out/s2n/err.txt | tail -1 > gpurun_out/s2n/gloo2_ahead.json
python - <<'PY'
import json
g=json.load(open("gpurun_out/s2n/gloo2_ahead.json")); c=g["config"]
print("n_gpus", g["n_gpus"], "ms", g["ms_per_step"], "ahead", c["backbone_ahead"], "voxels/batch", c["batch_voxels"], "multi", c["multi_gpu"])
PY
tail -3 gpurun_out/s2n/err.txt
