#!/bin/bash
# same-box A/B of the bench step: the in-tree library against tools/_ab/<name>.so      bash tools/ab_lib.sh liba4r_x.so [rounds]
for i in $(seq ${2:-3}); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-tree ', d['ms_per_step'])"
  A4R_LIB_PATH=tools/_ab/$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"
done
