#!/bin/bash
# A variant build of the library for A/B runs: tools/build_variant.sh <name> "<extra hipcc flags>" [files...]
#   -> asr_amd/libasrk_<name>.so   (objects in asr_amd/csrc/build/var_<name>; files not listed reuse build/*.o of the product)
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../asr_amd/csrc"
make -s -j8 > /dev/null
mkdir -p build/var_$name
objs=""
for f in asrk_*.hip; do
  o=build/${f%.hip}.o
  for g in "$@"; do
    if [ "$g" = "$f" ]; then
      o=build/var_$name/${f%.hip}.o
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags $( [ $f = asrk_logmel.hip ] && echo -fno-slp-vectorize ) -c $f -o $o &
    fi
  done
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libasrk_$name.so $objs
echo built asr_amd/libasrk_$name.so
