#!/bin/bash
# GPU box: batch sweeps of the image configs and the real-shaped text lines at larger batches    bash tools/_ab/r6_sweeps.sh <tag>
TAG=${1:-r06_f}
: > gpurun_out/${TAG}_mae_batch_sweep.jsonl
for dt in bf16 fp8; do for B in 8 32 64; do
  python bench.py --workload mae_compacter --dtype $dt --batch $B --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_mae_batch_sweep.jsonl
done; done
: > gpurun_out/${TAG}_vit_batch_sweep.jsonl
for dt in bf16 fp8; do for B in 8 16 32; do
  python bench.py --workload vit_lora --dtype $dt --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_vit_batch_sweep.jsonl
done; done
: > gpurun_out/${TAG}_real_shaped.jsonl
for B in 32 128 512; do
  python bench.py --real-shaped --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_real_shaped.jsonl
  python bench.py --real-shaped --workload roberta_pfeiffer_cpc --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/${TAG}_real_shaped.jsonl
done
for f in gpurun_out/${TAG}_mae_batch_sweep.jsonl gpurun_out/${TAG}_vit_batch_sweep.jsonl gpurun_out/${TAG}_real_shaped.jsonl; do echo $f; python - $f <<'EOP'
import json,sys
for l in open(sys.argv[1]):
    try:
        d=json.loads(l); r=d.get('roofline') or {}
        print(d['config']['workload'][:40], d['dtype'], 'B', d['config']['users_per_gpu'], d['ms_per_step'], 'ms', d['value'], 'user-seq/s', 'family frac', r.get('frac'), 'step frac', r.get('step_frac_of_peak'))
    except Exception as e: print('ERR', e, l[:80])
EOP
done
