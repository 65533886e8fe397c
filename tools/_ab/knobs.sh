run() { env $1 $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --workload mae_compacter --dtype $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mae $3 $1 $2', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
run A4R_X=0 A4R_Y=0 fp8
run A4R_GEMM_TAIL=7 A4R_TN2_WGS=192 fp8
run A4R_GEMM_TAIL=7 A4R_TN2_WGS=128 fp8
run A4R_GEMM_TAIL=5 A4R_TN2_WGS=192 fp8
run A4R_X=0 A4R_Y=0 bf16
run A4R_GEMM_TAIL=7 A4R_TN2_WGS=192 bf16
done
