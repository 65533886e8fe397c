# adapter backward prologue: Wd image by LDS-DMA + parameters written behind every request (in-tree) against the previous form (liba4r_ad_old.so)
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "adapter_ln" 2>&1 | tail -1
A4R_LIB_PATH=tools/_ab/liba4r_adstamp4.so timeout 300 python tools/adapter_launch_timeline.py 40448 bwd 2>&1 | grep -v "launches\|amdgpu\|fused"
for i in 1 2 3; do
  echo "new"; timeout 300 python tools/adapter_bench.py 2>&1 | grep "bwd fused"
  echo "old"; A4R_LIB_PATH=tools/_ab/liba4r_ad_old.so timeout 300 python tools/adapter_bench.py 2>&1 | grep "bwd fused"
done
echo "M=16896 new"; timeout 300 python tools/adapter_bench.py 16896 2>&1 | grep "fused (step)"
echo "M=16896 old"; A4R_LIB_PATH=tools/_ab/liba4r_ad_old.so timeout 300 python tools/adapter_bench.py 16896 2>&1 | grep "fused (step)"
