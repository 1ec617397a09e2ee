set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "variants_match" > gpurun_out/r04_call9_tests.txt 2>&1
tail -12 gpurun_out/r04_call9_tests.txt
SH=2:32:32,2:64:32,4:64:64,8:64:64,2:32:32:down,1:64:64:up,2:96:96:up
rm -f gpurun_out/r04_c9_lds.txt
for V in 0,0,0 32,6,0 0,0,0 32,6,0; do
timeout 300 env PP_CONV_VARIANT=$V python profiles/conv_one.py 64 $SH 5 2>&1 | grep "ts=" | sed "s/^/$V: /" >> gpurun_out/r04_c9_lds.txt
done
cat gpurun_out/r04_c9_lds.txt
timeout 600 python profiles/host_profile.py 1250000 > gpurun_out/r04_shard_host_profile.txt 2>&1; head -50 gpurun_out/r04_shard_host_profile.txt | cut -c1-180
