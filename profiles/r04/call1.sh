set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/microbench/gather_forms.hip -o /tmp/gf 2> /dev/null && /tmp/gf > gpurun_out/r04_gather_forms.txt 2>&1
python profiles/conv_one.py 64 1:16:16,1:64:16,1:16:32,2:32:32,2:96:32,4:48:48,4:128:48,8:64:64,16:80:80,1:64:64:up,2:96:96:up,2:16:16:down,4:32:32:down 5 > gpurun_out/r04_conv_base.txt 2>&1
bash profiles/clock_probe.sh base 64 1:16:16,2:32:32,4:48:48,8:64:64 > /dev/null 2>&1
tail -5 gpurun_out/r04_gather_forms.txt; tail -16 gpurun_out/r04_conv_base.txt; head -30 gpurun_out/clock_base.md
