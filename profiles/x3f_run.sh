# round 6: k_spconv_x3f on 48 / 80 / 112-channel inputs (odd number of 16-channel slots): parity, layer A/B, bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2d
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "x3 or spconv or transposed_map" 2>&1 | tail -4 > gpurun_out/s2d/tests.txt
cat gpurun_out/s2d/tests.txt
SH="4:48:48,16:80:80,4:48:64,8:48:48:down"
for v in 1 0 1 0; do echo "== PP_CONV_X3F=$v"; PP_CONV_X3F=$v python profiles/conv_one.py 64 $SH 5 2>/dev/null | grep -v amdgpu.ids; done > gpurun_out/s2d/conv_one_odd.txt 2>&1
cat gpurun_out/s2d/conv_one_odd.txt
