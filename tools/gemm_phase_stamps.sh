#!/bin/bash
# builds tools/_ab/liba4r_pst<S>.so (-DA4R_PHASE_STAMP -DA4R_DMA_SCHED=<S>) when missing and prints the phase tables   (SCHEDS="0 1" by default)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
for sc in ${SCHEDS:-1 0}; do
  so=tools/_ab/liba4r_pst$sc.so
  if [ ! -f $so ]; then
    ( cd adapter4rec_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DA4R_PHASE_STAMP -DA4R_DMA_SCHED=$sc -c a4r_gemm256.hip -o /tmp/g256_pst$sc.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../$so /tmp/g256_pst$sc.o $(ls *.o | grep -v a4r_gemm256.o) )
  fi
done
if [ -n "$BUILD_ONLY" ]; then exit 0; fi
for sc in ${SCHEDS:-1 0}; do
  for shape in "40448 768 3072" "40448 3072 768"; do
    echo "== A4R_DMA_SCHED=$sc"
    A4R_LIB_PATH=tools/_ab/liba4r_pst$sc.so python tools/gemm_phase_stamps.py $shape
  done
done
