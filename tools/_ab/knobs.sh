run() { env $1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'])"; }
for i in 1 2; do
for kv in A4R_X=0 A4R_GEMM_BAND_LONG=2 A4R_GEMM_BAND_LONG=4 A4R_TN2_WGS=256 A4R_TN2_WGS=512 A4R_TN2_WGS=768; do run $kv ""; done
done
