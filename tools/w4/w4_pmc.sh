#!/bin/bash
# counter passes over one GEMM shape for the eight-wave (8) and four-wave (9) kernels: fabric reads, L2 hits / misses, MFMA busy, GUI active
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
N=${N:-768}; K=${K:-3072}
for v in 8 9; do
  for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE"; do
    d=$OUT/pmc_tmp; rm -rf $d
    rocprofv3 --pmc $pass --kernel-trace -d $d -o p --output-format csv -- python3 $ROOT/tools/w4/w4_one.py $v $N $K > /dev/null 2>&1
    python3 - "$d" "$v" "$pass" <<'PY'
import csv, glob, os, sys, collections
d, v, p = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        if 'gemm_nt_256' not in r['Kernel_Name']:
            continue
        a = acc[(r['Kernel_Name'][:60], r['Counter_Name'])]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, s) in sorted(acc.items()):
    print(f'variant {v} {c:28s} per launch {s / n:16.1f}   ({n} launches) {k}')
PY
  done
done
