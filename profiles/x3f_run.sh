# round 6: k_spconv_x3f v3 (row offsets broadcast by DPP quad_perm instead of ds_bpermute; the next lines requested after the first
# tile's split instead of behind a wait for the fragment reads) against v2 (profiles/abl/libx3_f_base.so), same box, alternating
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/s2e
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or spconv or transposed_map" 2>&1 | tail -3 > gpurun_out/s2e/tests.txt
cat gpurun_out/s2e/tests.txt
SH="4:64:64,2:32:32,2:96:32,1:64:16,8:64:64,4:128:48,2:32:32:down,1:64:64:up"
for v in new f_base new f_base; do echo "== $v"; if [ $v = new ]; then python profiles/conv_one.py 64 $SH 5 2>/dev/null | grep -v amdgpu.ids; else PP_HIP_LIB=$GRAFT_REPO_ROOT/profiles/abl/libx3_$v.so python profiles/conv_one.py 64 $SH 5 2>/dev/null | grep -v amdgpu.ids; fi; done > gpurun_out/s2e/conv_one_v3.txt 2>&1
cat gpurun_out/s2e/conv_one_v3.txt
