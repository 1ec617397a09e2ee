set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "variants_match or spconv" > gpurun_out/r04_call5_tests.txt 2>&1
tail -6 gpurun_out/r04_call5_tests.txt
SH=1:16:16,1:64:16,1:16:32,2:32:32,2:96:32,4:48:48,4:128:48,8:64:64,16:80:80,1:64:64:up,2:96:96:up,2:16:16:down,4:32:32:down
rm -f gpurun_out/r04_call5_ab.txt
for rep in 1 2; do
python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/default$rep: /" >> gpurun_out/r04_call5_ab.txt
PP_CONV_VARIANT=0,5,0 python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/ring$rep:    /" >> gpurun_out/r04_call5_ab.txt
done
PP_CONV_VARIANT=32,5,0 python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/ring32:   /" >> gpurun_out/r04_call5_ab.txt
PP_CONV_VARIANT=64,5,0 python profiles/conv_one.py 64 $SH 5 2>&1 | grep -v amdgpu.ids | sed "s/^/ring64:   /" >> gpurun_out/r04_call5_ab.txt
cat gpurun_out/r04_call5_ab.txt
