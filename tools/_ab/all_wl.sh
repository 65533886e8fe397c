for wl in "bert_houlsby bf16" "bert_houlsby fp8" "roberta_pfeiffer_cpc bf16" "roberta_pfeiffer_cpc fp8" "vit_lora bf16" "vit_lora fp8" "mae_compacter bf16" "mae_compacter fp8"; do
  set -- $wl
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --workload $1 --dtype $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'], d['value'])"
done
