# same-box cost of the residual-stream forms on the headline step            bash tools/_ab/residual_ab.sh [rounds]
for i in $(seq ${1:-3}); do
  for rd in bf16 bf20 bf24; do
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --residual-dtype $rd 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rd', d['ms_per_step'], d['value'])"
  done
done
