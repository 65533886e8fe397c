#!/usr/bin/env python
"""How does a4r_gemm_nt compare with the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the step's shapes?
Measurement only (the product never calls the vendor GEMM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

def t_us(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device('cuda:0')
for M in (40448, 66304):
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev)
        bb = bias.bfloat16()
        ta = t_us(lambda: L.gemm_nt(A, B, C, bias=bias))
        tv = t_us(lambda: torch.nn.functional.linear(A, B, bb))
        f = 2.0 * M * N * K
        print(f'M={M} N={N} K={K}: a4r {f/ta/1e6:7.1f} TF/s ({ta:.1f} us)   vendor {f/tv/1e6:7.1f} TF/s ({tv:.1f} us)')
