#!/bin/bash
# A/B of builds of the library on ONE box (rank_scaling: the sweep kernel alone): tools/ab_rank.sh NAME... ; "new" = the in-tree build,
# other names = kge-keras_amd/csrc/variants/libkge_hip_NAME.so (tools/build_variant.sh)
for rep in 1 2; do
for v in "$@"; do
  if [ $v = new ]; then unset KGE_HIP_LIB; else export KGE_HIP_LIB=$PWD/kge-keras_amd/csrc/variants/libkge_hip_$v.so; fi
  python tools/rank_scaling.py 2>/dev/null | sed -n 4,5p | sed "s/^/$v /"
  python tools/rank_scaling.py 40943 400 ComplEx 2>/dev/null | sed -n 3,3p | sed "s/^/$v /"
done; done
