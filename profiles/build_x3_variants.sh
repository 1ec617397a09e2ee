#!/bin/bash
# A/B builds of the split-bf16 convolution kernel: bash profiles/build_x3_variants.sh "<tag> <flags>" ...  ->  profiles/abl/libx3_<tag>.so
# (run in the build container; the libraries travel to the GPU box with the snapshot; select one with PP_HIP_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/panopticsegforlargescalepointcloud_amd/csrc
mkdir -p $R/profiles/abl
make -C $C -j8 ARCH=gfx950 > /dev/null
OBJS=$(ls $C/*.o | grep -v pp_spconv3.o)
for spec in "$@"; do
  tag=${spec%% *}; flags=${spec#* }
  [ "$flags" = "$tag" ] && flags=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $C/pp_spconv3.hip -o /tmp/x3_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/abl/libx3_$tag.so $OBJS /tmp/x3_$tag.o
  echo "built libx3_$tag.so ($flags)"
done
