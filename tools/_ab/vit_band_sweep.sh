for i in 1 2; do
for b in 3 2 4 6; do
  A4R_GEMM_BAND_LONG=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit band_long=$b', d['ms_per_step'], d['value'])"
done
done
for s in 30 0 50; do
  A4R_GEMM_STAGGER=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit stagger=$s', d['ms_per_step'], d['value'])"
done
