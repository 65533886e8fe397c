#!/bin/bash
# same-box A/B of A4R_WGRAD_STREAM (adapter weight gradients on a side stream) on the workloads whose N = 768 launches leave CUs idle
for wl in "mae_compacter bf16" "mae_compacter fp8" "vit_lora bf16" "roberta_pfeiffer_cpc bf16"; do
  set -- $wl
  for i in 1 2 3; do
    for v in 0 1; do
      A4R_WGRAD_STREAM=$v python bench.py --workload $1 --dtype $2 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 WGRAD_STREAM=$v', d['ms_per_step'], d['value'])"
    done
  done
done
