# adapter forward, register form: timing-only builds (wrong results on purpose): 8 no row loads in the loop, 16 no stores, 24 neither
for i in 1 2; do
for l in "" liba4r_adabl8.so liba4r_adabl16.so liba4r_adabl24.so; do
  echo "lib=${l:-in-tree}"; A4R_AD_RING=0 A4R_LIB_PATH=${l:+tools/_ab/$l} timeout 300 python tools/adapter_bench.py 2>&1 | grep "fused (step)"
done
done
