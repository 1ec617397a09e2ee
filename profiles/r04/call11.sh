set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SH=1:64:16,1:32:16,2:96:32,2:64:32,4:128:48
rm -f gpurun_out/r04_c11_lds.txt
for V in 0,0,0 32,6,0 32,1,0 0,0,0 32,6,0; do
timeout 300 env PP_CONV_VARIANT=$V python profiles/conv_one.py 64 $SH 5 2>&1 | grep "ts=" | sed "s/^/$V: /" >> gpurun_out/r04_c11_lds.txt
done
cat gpurun_out/r04_c11_lds.txt
timeout 600 python profiles/host_profile.py 1250000 > gpurun_out/r04_shard_host_profile.txt 2>&1; grep -n "was called by" -A12 gpurun_out/r04_shard_host_profile.txt | cut -c1-220 | head -70
