# round 6: k_spconv_x3f (full-line gathers through LDS) against k_spconv_x3 (PP_CONV_X3F=0), layer by layer on the 64-tile scene's maps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x3f
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or spconv" 2>&1 | tail -5 > gpurun_out/x3f/tests.txt
cat gpurun_out/x3f/tests.txt
SH="4:48:48,8:64:64,4:128:48,1:64:64:up,8:64:64:down,4:64:64,2:32:32,2:96:32,2:64:32,16:80:80,4:96:96:up,2:32:32:down,4:32:64,16:160:64"
for v in 1 0 1 0; do echo "== PP_CONV_X3F=$v"; PP_CONV_X3F=$v python profiles/conv_one.py 64 $SH 5 2>/dev/null | grep -v amdgpu.ids; done > gpurun_out/x3f/conv_one.txt 2>&1
cat gpurun_out/x3f/conv_one.txt
