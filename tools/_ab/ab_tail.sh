for i in 1 2 3; do
  for mf in 64 0; do
  A4R_GEMM_TAIL_MINFRAC=$mf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit minfrac $mf', d['ms_per_step'], d['value'])"
  done
done
for mf in 64 0 64 0; do
  A4R_GEMM_TAIL_MINFRAC=$mf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora --dtype fp8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit fp8 minfrac $mf', d['ms_per_step'], d['value'])"
done
