for i in 1 2 3; do
  for v in 1 0; do
  A4R_GEMM_BAND_TAIL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit band_tail=$v', d['ms_per_step'], d['value'])"
  done
done
for v in 1 0 1 0; do
A4R_GEMM_BAND_TAIL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora --dtype fp8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit fp8 band_tail=$v', d['ms_per_step'], d['value'])"
done
