for i in 1 2 3; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'])"
  A4R_LIB_PATH=tools/_ab/liba4r_old.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['value'])"
done
