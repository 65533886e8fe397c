# GEMM epilogue: write-through (sc1 / sc0 sc1) output stores against the write-back default: per-form times, then the step, alternating
for l in "" liba4r_wt2048.so liba4r_wt4096.so; do
  echo "== lib=${l:-in-tree}"; A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 600 python tools/gemm_forms.py 40448 2>&1 | grep -v amdgpu
done
bash tools/ab_lib.sh liba4r_wt2048.so 3
bash tools/ab_lib.sh liba4r_wt4096.so 2
