#!/bin/bash
# Where does k_spconv_fwd3 spend its time?  Builds four ablated copies of the library (F3_ABLATE=1..4 in pp_spconv2.hip:
# no step loop / no MFMAs / no feature gathers / no weight loads) into profiles/abl/ and times one layer shape with each.
#   build here (no GPU needed):  bash profiles/ablate_conv.sh build
#   on the GPU box:              bash profiles/ablate_conv.sh run <n_tiles> <ts> <cin> <cout>
# (profiles/abl/ is listed in .gitignore and .gpurunignore: drop it from .gpurunignore to ship the builds to the box)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/panopticsegforlargescalepointcloud_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/profiles/abl
  for n in 1 2 3 4; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DF3_ABLATE=$n -c $C/pp_spconv2.hip -o $R/profiles/abl/spconv2_$n.o
    objs=$(ls $C/*.o | grep -v pp_spconv2.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/abl/libpanoptic_abl$n.so $objs $R/profiles/abl/spconv2_$n.o
  done
  rm -f $R/profiles/abl/*.o
else
  shift
  python $R/profiles/conv_one.py "$@" dense 10
  for n in 1 2 3 4; do
    PP_HIP_LIB=$R/profiles/abl/libpanoptic_abl$n.so python $R/profiles/conv_one.py "$@" dense 10 | sed "s/^/ablate $n: /"
  done
fi
