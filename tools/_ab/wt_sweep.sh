# A4R_GEMM_WT_ROUNDS10: write-through stores on a workgroup's last tile when the launch has at most that many tenths of a round of tiles; 0 = never
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2> /tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" || tail -3 /tmp/err.txt; }
for rep in 1 2; do
for wl in "bert_houlsby" "mae_compacter" "mae_compacter --dtype fp8" "roberta_pfeiffer_cpc" "vit_lora"; do
  for t in 0 12 25 35 60; do
    echo -n "$wl rounds10=$t "; A4R_GEMM_WT_ROUNDS10=$t run --workload $wl
  done
done
done
