for i in 1 2 3; do
  for f in 1 0; do
  A4R_LORA_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit fused=$f', d['ms_per_step'], d['value'])"
  done
done
for f in 1 0; do
A4R_LORA_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload vit_lora --dtype fp8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit fp8 fused=$f', d['ms_per_step'], d['value'])"
done
