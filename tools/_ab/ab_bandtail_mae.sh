for i in 1 2 3; do
  for v in 1 0; do
  A4R_GEMM_BAND_TAIL=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --workload mae_compacter 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mae band_tail=$v', d['ms_per_step'], d['value'])"
  done
done
