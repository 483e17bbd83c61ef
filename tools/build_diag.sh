#!/bin/bash
# Diagnostic library with in-kernel stamps (never loaded by the product): asr_amd/libasrk_diag.so
set -e
cd "$(dirname "$0")/../asr_amd/csrc"
mkdir -p build/diag
for f in asrk_*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DASRK_STAMPS -c $f -o build/diag/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libasrk_diag.so build/diag/*.o
