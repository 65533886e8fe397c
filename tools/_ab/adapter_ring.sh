# adapter forward: LDS-DMA ring (A4R_AD_RING=1, default) against the register form (=0): tests, then timing, alternating
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "adapter_ln" 2>&1 | tail -3
A4R_AD_RING=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "adapter_ln_fwd" 2>&1 | tail -1
for i in 1 2 3; do
  echo "ring=1"; A4R_AD_RING=1 timeout 300 python tools/adapter_bench.py 2>&1 | grep "fused"
  echo "ring=0"; A4R_AD_RING=0 timeout 300 python tools/adapter_bench.py 2>&1 | grep "fused"
done
echo "M=66304 ring=1"; A4R_AD_RING=1 timeout 300 python tools/adapter_bench.py 66304 2>&1 | grep "fused (step)"
echo "M=66304 ring=0"; A4R_AD_RING=0 timeout 300 python tools/adapter_bench.py 66304 2>&1 | grep "fused (step)"
