# write-through stores on a workgroup's last tile (in-tree) against write-back everywhere (tools/_ab/liba4r_wb.so)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -1
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2> /tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'])" || tail -3 /tmp/err.txt; }
for i in 1 2 3; do
  echo -n "wt-last headline "; run
  echo -n "write-back headline "; A4R_LIB_PATH=tools/_ab/liba4r_wb.so run
done
for wl in "mae_compacter" "mae_compacter --dtype fp8" "roberta_pfeiffer_cpc" "vit_lora"; do for i in 1 2; do
  echo -n "wt-last $wl "; run --workload $wl
  echo -n "write-back $wl "; A4R_LIB_PATH=tools/_ab/liba4r_wb.so run --workload $wl
done; done
for l in "" liba4r_wb.so; do
  echo "== M=16896 lib=${l:-in-tree}"; A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 600 python tools/gemm_forms.py 16896 2>&1 | grep -v amdgpu | cut -c1-75
done
for l in "" liba4r_wb.so; do
  echo "== M=40448 lib=${l:-in-tree}"; A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 600 python tools/gemm_forms.py 40448 2>&1 | grep -v amdgpu | cut -c1-75
done
