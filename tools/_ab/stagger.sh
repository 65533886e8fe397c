for i in 1 2; do
for s in 30 50 65 80 100; do
  A4R_GEMM_STAGGER=$s python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bert stagger=$s', d['ms_per_step'])"
done
done
for s in 30 50 65 80; do
  A4R_GEMM_STAGGER=$s python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --workload roberta_pfeiffer_cpc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('roberta stagger=$s', d['ms_per_step'])"
  A4R_GEMM_STAGGER=$s python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --workload mae_compacter 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mae stagger=$s', d['ms_per_step'])"
  A4R_GEMM_STAGGER=$s python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --dtype fp8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bert fp8 stagger=$s', d['ms_per_step'])"
done
