#!/usr/bin/env python
"""a4r_gemm_nt variants (4 = eight-wave 256 tile, 5 = four-wave stream kernel) against the vendor library on the step's shapes.
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
variants = [int(v) for v in os.environ.get('VARIANTS', '4,5').split(',')]
for M in (40448,):
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304), (768, 64)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev)
        R1 = torch.randn(M, N, device=dev).bfloat16()
        bb = bias.bfloat16()
        f = 2.0 * M * N * K
        line = f'M={M} N={N} K={K}:'
        ref = None
        for v in variants:
            L.gemm_variant(v)
            t = t_us(lambda: L.gemm_nt(A, B, C, bias=bias))
            if ref is None: ref = C.clone()
            eq = ' eq' if torch.equal(ref, C) else ' DIFF'
            td = t_us(lambda: L.gemm_nt(A, B, C, bias=bias, R1=R1, drop_p=0.1, drop_site=1, drop_seed=5, drop_first=True))
            line += f'  v{v} {f/t/1e6:7.1f} TF/s ({t:.1f} us; +R1+dropout {td:.1f} us){eq}'
        L.gemm_variant(2)
        tv = t_us(lambda: torch.nn.functional.linear(A, B, bb))
        print(line + f'   vendor {f/tv/1e6:7.1f} TF/s ({tv:.1f} us)', flush=True)
