# fragment-ordered adapter weights (ABI 408): tests, then the step with A4R_ADAPTER_FRAG=1 (default) / 0, alternating, three workloads
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "adapter_ln" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_cv.py tests/test_dropout_parity_gpu.py tests/test_torch_ops.py -q -x 2>&1 | tail -2
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'])"; }
for i in 1 2 3; do
  echo -n "frag=1 headline "; A4R_ADAPTER_FRAG=1 run
  echo -n "frag=0 headline "; A4R_ADAPTER_FRAG=0 run
done
for wl in mae_compacter roberta_pfeiffer_cpc vit_lora; do for i in 1 2; do
  echo -n "frag=1 $wl "; A4R_ADAPTER_FRAG=1 run --workload $wl
  echo -n "frag=0 $wl "; A4R_ADAPTER_FRAG=0 run --workload $wl
done; done
echo -n "frag=1 mae fp8 "; A4R_ADAPTER_FRAG=1 run --workload mae_compacter --dtype fp8
echo -n "frag=0 mae fp8 "; A4R_ADAPTER_FRAG=0 run --workload mae_compacter --dtype fp8
