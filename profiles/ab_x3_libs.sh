#!/bin/bash
# conv_one.py over a shape list for every A/B library: bash profiles/ab_x3_libs.sh <out> "<shapes>" <lib tags...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; SH=$2; shift 2
: > $OUT
for t in "$@"; do
  echo "== $t" >> $OUT
  PP_HIP_LIB=$R/profiles/abl/libx3_$t.so python $R/profiles/conv_one.py 64 $SH 5 2>/dev/null | grep -v amdgpu.ids >> $OUT
done
cat $OUT
