#!/bin/bash
# Where does k_spconv_fwd3 spend its time?  Builds four ablated copies of the library (F3_ABLATE=1..4 in pp_spconv2.hip:
# no step loop / no MFMAs / no feature gathers / no weight loads) into profiles/abl/ and times one layer shape with each.
#   build here (no GPU needed):  bash profiles/ablate_conv.sh build abl1 -DF3_ABLATE=1     (any name / flags)
#   on the GPU box:              bash profiles/ablate_conv.sh run <n_tiles> <ts:cin:cout[:kind],...>
# (profiles/abl/ is listed in .gitignore and .gpurunignore: drop it from .gpurunignore to ship the builds to the box)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/panopticsegforlargescalepointcloud_amd/csrc
if [ "$1" = build ]; then      # build <name> <extra hipcc flags...>: one A/B copy of the library with pp_spconv2.hip rebuilt
  name=$2; shift 2
  mkdir -p $R/profiles/abl
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $C/pp_spconv2.hip -o $R/profiles/abl/spconv2_$name.o
  objs=$(ls $C/*.o | grep -v pp_spconv2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/abl/libpanoptic_$name.so $objs $R/profiles/abl/spconv2_$name.o
  rm -f $R/profiles/abl/*.o
else                           # run <n_tiles> <shapes>: every copy in profiles/abl/ next to the product library
  shift
  python $R/profiles/conv_one.py "$@" 10 | sed "s/^/product: /"
  for lib in $R/profiles/abl/libpanoptic_*.so; do
    PP_HIP_LIB=$lib python $R/profiles/conv_one.py "$@" 10 | sed "s/^/$(basename $lib .so | sed s/libpanoptic_//): /"
  done
fi
