# write-through (sc1) epilogue stores on SHORT launches: M = 16 896 (the image tower under ViT-MAE) and M = 40 448, alternating
for i in 1 2 3; do
  for l in "" liba4r_wt2048.so; do
    echo "== M=16896 lib=${l:-in-tree}"; A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 600 python tools/gemm_forms.py 16896 2>&1 | grep -v amdgpu | cut -c1-75
  done
done
for i in 1 2; do
  for l in "" liba4r_wt2048.so; do
    echo "== M=40448 lib=${l:-in-tree}"; A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 600 python tools/gemm_forms.py 40448 2>&1 | grep -v amdgpu | grep "attn-out" | cut -c1-75
  done
done
