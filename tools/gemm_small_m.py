#!/usr/bin/env python
"""The 256-tile ring kernel against the 128-tile kernel (a4r_gemm_variant 2 / 1) at ViT-MAE's 16 896 rows (66 row panels: under one round of 256-wide tiles
for N = 768) and at the CLS-only last layer's 1 408 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device('cuda:0')
for M in (16896, 1408):
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        out = []
        for v in (2, 1):
            L.gemm_variant(v)
            out.append(t_us(lambda: L.gemm_nt(A, B, C)))
        L.gemm_variant(2)
        print(f'M={M} N={N} K={K}: default {out[0]:6.1f} us   128-tile kernel {out[1]:6.1f} us   ({2.0 * M * N * K / out[0] / 1e6:5.0f} / {2.0 * M * N * K / out[1] / 1e6:5.0f} TF/s)')
