#!/bin/bash
# A/B builds of libkge_hip.so: tools/build_variant.sh NAME "-DKGE_PULL_DIST2=1 ..." [file.hip ...]
# recompiles the named translation units (default: pull.hip pullstep.hip) with the extra flags and links them with the objects of the
# default build into kge-keras_amd/csrc/variants/libkge_hip_NAME.so (run with KGE_HIP_LIB=<that path>).
set -e
cd "$(dirname "$0")/../kge-keras_amd/csrc"
name=$1; flags=$2; shift 2 || true
files=${@:-pull.hip pullstep.hip}
make -j8 >/dev/null
mkdir -p variants/$name
objs=""
for o in *.o; do
  f=${o%.o}.hip
  if [[ " $files " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $flags -c $f -o variants/$name/$o &
    objs="$objs variants/$name/$o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -L/opt/rocm/lib -lrocblas -ldl -Wl,-rpath,/opt/rocm/lib -o variants/libkge_hip_$name.so
echo variants/libkge_hip_$name.so
