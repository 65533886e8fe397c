#!/bin/bash
# same-box A/B of the bench step between two environments:  bash tools/_ab/ab_env_step.sh "A4R_GEMM_W4=0" "A4R_GEMM_W4=1" [rounds] [extra bench args]
A="$1"; B="$2"; R=${3:-3}; shift 3
for i in $(seq $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['ms_per_step'], d['value'])"
  done
done
