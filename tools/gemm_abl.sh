#!/bin/bash
# Timing-only ablation builds of the 256-tile GEMM (results are WRONG in these builds; they tell where the K loop's time goes):
#   bash tools/gemm_abl.sh            (on the GPU box; builds tools/_ab/liba4r_abl<N>.so for each N and runs tools/gemm_forms.py plain rows)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
for abl in ${ABLS:-0 1 4 5 8 16 2 3 7}; do
  so=tools/_ab/liba4r_abl$abl.so
  if [ ! -f $so ]; then
    ( cd adapter4rec_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DA4R_ABL=$abl -c a4r_gemm256.hip -o /tmp/g256_$abl.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../$so /tmp/g256_$abl.o $(ls *.o | grep -v a4r_gemm256.o) )
  fi
done
if [ -n "$BUILD_ONLY" ]; then exit 0; fi
for abl in ${ABLS:-0 1 4 5 8 16 2 3 7}; do
  echo "ABL=$abl"
  A4R_LIB_PATH=tools/_ab/liba4r_abl$abl.so python tools/gemm_forms.py 40448 plainonly 2>&1 | grep -v vendor_only
done
