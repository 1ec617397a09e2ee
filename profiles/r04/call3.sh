set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_fullsize_gpu.py -m gpu -x -q -k "conv or map or shortcut or transposed or split" > gpurun_out/r04_call3_tests.txt 2>&1
tail -8 gpurun_out/r04_call3_tests.txt
SH=1:16:16,1:64:16,1:16:32,2:32:32,2:96:32,4:48:48,4:128:48,8:64:64,16:80:80,1:64:64:up,2:96:96:up,2:16:16:down,4:32:32:down
rm -f gpurun_out/r04_call3_ab.txt
for rep in 1 2; do
PP_HIP_LIB=$PWD/profiles/abl/libpanoptic_base.so python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/base$rep: /" >> gpurun_out/r04_call3_ab.txt
python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/new$rep:  /" >> gpurun_out/r04_call3_ab.txt
done
cat gpurun_out/r04_call3_ab.txt
python -m pytest tests/test_edge_cases_gpu.py -m gpu -x -q -k "storage_swap or frozen_second or no_grad" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
