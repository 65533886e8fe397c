#!/bin/bash
# SURVEY 8(d) batch sweep of the headline workload + the other BASELINE workloads on the current code (one box):  bash tools/_ab/batch_sweep.sh <tag>
TAG=${1:-r05_d}; OUT=gpurun_out; mkdir -p $OUT
: > $OUT/${TAG}_batch_sweep.jsonl
for B in 8 32 64 128 256; do
  S=30; [ $B -ge 128 ] && S=12
  python bench.py --batch $B --steps $S --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/${TAG}_batch_sweep.jsonl
done
for wl in mae_compacter vit_lora roberta_pfeiffer_cpc; do
  for dt in bf16 fp8; do
    python bench.py --workload $wl --dtype $dt --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_${wl}_${dt}.json
  done
done
python bench.py --dtype fp8 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_bert_houlsby_fp8.json
python - <<'PY'
import json, glob, os
tag = os.environ.get('TAG', 'r05_d')
for l in open(f'gpurun_out/{tag}_batch_sweep.jsonl'):
    d = json.loads(l); r = d.get('roofline') or {}
    print('B', d['config']['users_per_gpu'], d['value'], 'user-seq/s', d['ms_per_step'], 'ms  step_frac', r.get('step_frac_of_peak'), 'gemm frac', r.get('frac'))
for f in sorted(glob.glob(f'gpurun_out/{tag}_bench_*.json')):
    d = json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'])
PY
