#!/bin/bash
# one-launch long-attention backward (A4R_ATTN_BWD_FUSED=1) against the two launches (0): the kernels alone, then the ViT-B/16 + LoRA and ViT-MAE + Compacter steps
for v in 0 1 0 1; do echo "FUSED=$v"; A4R_ATTN_BWD_FUSED=$v python tools/attn_bench.py 2>&1 | grep "S="; done
for wl in "vit_lora bf16" "mae_compacter bf16" "mae_compacter fp8"; do
  set -- $wl
  for i in 1 2 3; do
    for v in 0 1; do
      A4R_ATTN_BWD_FUSED=$v python bench.py --workload $1 --dtype $2 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 ATTN_BWD_FUSED=$v', d['ms_per_step'], d['value'])"
    done
  done
done
