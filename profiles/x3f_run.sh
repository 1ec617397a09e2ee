# round 6: --backbone-ahead auto: the default line (9.8 M voxels: off), configs[1] (2 M voxels: on), one rank's share (on)
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2m
P='
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]
print("ms_per_step %.2f value %.3e frac %.4f single %s ahead %s mode %s voxels %s checks %s"%(d["ms_per_step"], d["value"], r["frac"], c["single_scene_ms"], c["backbone_ahead"], c["backbone_ahead_mode"], c["batch_voxels"], (c.get("checks") or {}).get("all")))
'
echo "== default"; timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
for v in auto off auto off; do echo "== C2 (--points 2000000 --grid 4) --backbone-ahead $v"; timeout 600 python bench.py --points 2000000 --grid 4 --steps 20 --warmup 3 --no-cpu-baseline --backbone-ahead $v 2>/dev/null | tail -1 | python -c "$P"; done
echo "== share auto"; timeout 600 python bench.py --points 1250000 --grid 3 --steps 30 --warmup 4 --no-cpu-baseline --no-checks 2>/dev/null | tail -1 | python -c "$P"
