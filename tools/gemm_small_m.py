#!/usr/bin/env python
"""The 256-tile ring kernel (a4r_gemm_variant 4: forced) against the automatic choice (2) and the 128-tile kernel (1) on UNDER-FILLED shapes: few row
panels (8 users: 40; ViT-MAE: 66; the CLS-only last layer) -- where does the 256-tile kernel start to win?   usage: python tools/gemm_small_m.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

def t_us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device('cuda:0')
Ms = [int(x) for x in sys.argv[1:]] or [1536, 2560, 5120, 7680, 10240, 16896]
for M in Ms:
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        bias = torch.zeros(N, device=dev)
        out = {}
        for rnd in range(2):
            for v in (2, 4, 1):
                L.gemm_variant(v)
                out.setdefault(v, []).append(t_us(lambda: L.gemm_nt(A, B, C, bias=bias)))
        L.gemm_variant(2)
        fl = 2.0 * M * N * K
        print(f'M={M:6d} ({M // 256 * (N // 256):4d} tiles) N={N:4d} K={K:4d}: auto {min(out[2]):6.1f} us | 256-tile {min(out[4]):6.1f} us | 128-tile {min(out[1]):6.1f} us   ({fl / min(out[2]) / 1e6:5.0f} / {fl / min(out[4]) / 1e6:5.0f} / {fl / min(out[1]) / 1e6:5.0f} TF/s)', flush=True)
