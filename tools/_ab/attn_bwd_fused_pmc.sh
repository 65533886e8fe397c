#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the long attention backward as two launches (A4R_ATTN_BWD_FUSED=0) and as one (=1), ViT-MAE + Compacter step and ViT-B/16 + LoRA step
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in mae_compacter vit_lora; do
for v in 0 1; do
  export A4R_ATTN_BWD_FUSED=$v
  rm -rf $OUT/abf_f $OUT/abf_w
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/abf_f -o f --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $wl > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/abf_w -o w --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $wl > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/abf_f $OUT/abf_w > $OUT/abf_${wl}_$v.json
  python3 - $OUT/abf_${wl}_$v.json $wl $v <<'EOF'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict):
        for name, row in v.items():
            if 'attn_long' in str(name):
                print(sys.argv[2], 'FUSED=' + sys.argv[3], str(name)[:70], row)
EOF
done
done
rm -rf $OUT/abf_f $OUT/abf_w
