#!/bin/bash
# Same-box A/B of two builds of liba4r_hip.so on the default bench step: tools/ab_bench.sh <base.so> [rounds]
# (boxes differ by several % in wall time; only interleaved runs on one box compare)
base=${1:-tools/_ab/liba4r_hip_base.so}; rounds=${2:-3}
for i in $(seq $rounds); do
  for lib in "$base" ""; do
    A4R_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --steps 30 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-current}', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['all_gemm_tflops'])"
  done
done
