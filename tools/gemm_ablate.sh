#!/bin/bash
# Diagnostic: the stamps library (tools/build_diag.sh first) with parts of the persistent GEMM's K loop removed after the
# prologue (results are garbage, times are what is measured): asr_amd/libasrk_abl<N>.so for N = 1 (no LDS-DMA),
# 2 (no fragment reads), 3 (neither: MFMAs, waits and barriers only), 4 / 8 / 12 (X / W / both always re-read the first
# tile's slice 0: DMA issue and LDS traffic stay, the memory latency goes).  Run tools/gemm_pstamps.py with ASRK_DIAG_LIB set.
set -e
cd "$(dirname "$0")/../asr_amd/csrc"
for n in ${ABL:-1 2 3 4 8 12}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DASRK_DIAG -DASRK_STAMPS -DASRK_ABLATE=$n -c asrk_gemm256p.hip -o build/stamps/abl$n.o &
done
wait
for n in ${ABL:-1 2 3 4 8 12}; do
  objs=$(ls build/stamps/asrk_*.o | grep -v asrk_gemm256p.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libasrk_abl$n.so $objs build/stamps/abl$n.o
done
rm -f build/stamps/abl*.o
