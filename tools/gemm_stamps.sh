#!/bin/bash
# tools/gemm_stamps.sh [abl ...]: diagnostic builds (-DA4R_STAMP [-DA4R_ABL=n]) of the 256-tile GEMM + tools/gemm_stamps.py on each.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
for abl in ${@:-0}; do
  so=tools/_ab/liba4r_stamp$abl.so
  if [ ! -f $so ]; then
  ( cd adapter4rec_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DA4R_STAMP -DA4R_ABL=$abl -c a4r_gemm256.hip -o /tmp/g256_st$abl.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../$so /tmp/g256_st$abl.o $(ls *.o | grep -v a4r_gemm256.o) )
  fi
done
if [ -n "$BUILD_ONLY" ]; then exit 0; fi
for abl in ${@:-0}; do echo "ABL=$abl"; A4R_LIB_PATH=tools/_ab/liba4r_stamp$abl.so python tools/gemm_stamps.py; done
