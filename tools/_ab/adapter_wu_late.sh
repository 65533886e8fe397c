# adapter forward: up-projection fragments requested behind the first tile's rows (in-tree, A4R_AD_WU_LATE=1) against the old order (liba4r_wu0.so)
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "adapter_ln" 2>&1 | tail -1
for l in wu1_st3 wu0_st3; do echo "== $l"; A4R_LIB_PATH=tools/_ab/liba4r_$l.so timeout 300 python tools/adapter_launch_timeline.py 2>&1 | grep -v "fused\|launches\|amdgpu"; done
for i in 1 2 3; do
  echo "wu late"; timeout 300 python tools/adapter_bench.py 2>&1 | grep "fwd fused (step)"
  echo "wu first"; A4R_LIB_PATH=tools/_ab/liba4r_wu0.so timeout 300 python tools/adapter_bench.py 2>&1 | grep "fwd fused (step)"
done
