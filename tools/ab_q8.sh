for i in 1 2 3; do
for q in 1 0; do
A4R_Q8_DERIV=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q8=$q', d['ms_per_step'])"
done; done
