# the round's final artifact set (gpurun_out/<tag>_*): copy what is cited to profiles/          bash tools/_ab/final_profiles_r6.sh r06_z
TAG=${1:-r06_z}
bash tools/profile_step.sh $TAG pmc bert_houlsby bf16 > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_hbm_traffic.json profiles/ 2>/dev/null      # bench.py reads the newest traffic file for its roofline.traffic
python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
python bench.py --residual-dtype bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_full_residual_bf16.json 2>/dev/null
for wl in "bert_houlsby fp8" "roberta_pfeiffer_cpc bf16" "roberta_pfeiffer_cpc fp8" "vit_lora bf16" "vit_lora fp8" "mae_compacter bf16" "mae_compacter fp8"; do
  set -- $wl
  sfx=""; [ "$2" = "fp8" ] && sfx="_fp8"
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --workload $1 --dtype $2 > gpurun_out/${TAG}_bench_$1$sfx.json 2>/dev/null
done
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --workload bert_pretrain > gpurun_out/${TAG}_bench_bert_pretrain.json 2>/dev/null
bash tools/profile_step.sh ${TAG}_vit pmc vit_lora bf16 > /dev/null 2>&1
bash tools/profile_step.sh ${TAG}_mae nopmc mae_compacter fp8 > /dev/null 2>&1
bash tools/profile_step.sh ${TAG}_pretrain nopmc bert_pretrain bf16 > /dev/null 2>&1
python tools/attn_bench.py > gpurun_out/${TAG}_kernels.txt 2>&1
A4R_ATTN_BWD_ONEPASS=0 python tools/attn_bench.py 2>&1 | sed 's/^/two-launch backward (A4R_ATTN_BWD_ONEPASS=0): /' >> gpurun_out/${TAG}_kernels.txt
python tools/ln_bench.py >> gpurun_out/${TAG}_kernels.txt 2>&1
python tools/lora_bench.py >> gpurun_out/${TAG}_kernels.txt 2>&1
python tools/adapter_bench.py 2>&1 | tail -4 >> gpurun_out/${TAG}_kernels.txt
for f in gpurun_out/${TAG}_bench_*.json; do python - $f <<'EOP'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'])
except Exception as e: print(sys.argv[1], 'ERR', e)
EOP
done
python tools/eval_bench.py --json gpurun_out/${TAG}_eval_bench.json > /dev/null 2> gpurun_out/${TAG}_eval_bench.err
: > gpurun_out/${TAG}_bench_realistic.jsonl
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --short-titles >> gpurun_out/${TAG}_bench_realistic.jsonl 2>/dev/null
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --ragged-histories >> gpurun_out/${TAG}_bench_realistic.jsonl 2>/dev/null
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --short-titles --ragged-histories >> gpurun_out/${TAG}_bench_realistic.jsonl 2>/dev/null
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --real-shaped >> gpurun_out/${TAG}_bench_realistic.jsonl 2>/dev/null
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --workload roberta_pfeiffer_cpc --short-titles --ragged-histories >> gpurun_out/${TAG}_bench_realistic.jsonl 2>/dev/null
python tools/gemm_forms.py 40448 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_gemm_forms.txt
tail -c 1200 gpurun_out/${TAG}_bench_full.json
