#!/bin/bash
# same-box A/B of the bench step over one environment variable:   bash tools/ab_env.sh A4R_FUSE_BD 1 0 [rounds]
for i in $(seq ${4:-3}); do
  for v in $2 $3; do
    env $1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', d['ms_per_step'])"
  done
done
