#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps on top of the measurement switches (never loaded by the product or
# the tests): asr_amd/libasrk_stamps.so, read by tools/gemm_stamps.py and tools/gemm_pstamps.py.
set -e
cd "$(dirname "$0")/../asr_amd/csrc"
mkdir -p build/stamps
for f in asrk_*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DASRK_DIAG -DASRK_STAMPS $( [ $f = asrk_logmel.hip ] && echo -fno-slp-vectorize ) -c $f -o build/stamps/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libasrk_stamps.so build/stamps/*.o
