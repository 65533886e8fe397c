#!/usr/bin/env python
"""N = 64 GEMM (adapter down-projection shape): skinny64_kernel (variant 2) against the 128-row tile (variant 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L
def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dev = torch.device('cuda:0')
for M, K in ((40448, 768), (66304, 768), (40448, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(64, K, device=dev) * 0.05).bfloat16()
    C = torch.empty(M, 64, device=dev, dtype=torch.bfloat16); C2 = torch.empty_like(C); bias = torch.zeros(64, device=dev)
    line = f'M={M} K={K}:'
    for v in (1, 2):
        L.gemm_variant(v)
        t = t_us(lambda: L.gemm_nt(A, B, C, bias=bias, C2=C2, act=1))
        line += f'  variant {v}: {t:.1f} us ({(M * K * 2 + 2 * M * 128) / t / 1e6:.2f} TB/s)'
    L.gemm_variant(2)
    print(line, flush=True)
print('K = 64 (adapter up-projection shape, bias + two residuals): skinnyk_kernel (variant 2) against the 256-tile kernel (variant 4)')
for M, N in ((40448, 768), (66304, 768)):
    A = torch.randn(M, 64, device=dev).bfloat16(); B = (torch.randn(N, 64, device=dev) * 0.05).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev)
    R1 = torch.randn(M, N, device=dev).bfloat16(); R2 = torch.randn(M, N, device=dev).bfloat16()
    line = f'M={M} N={N}:'
    for v in (4, 2):
        L.gemm_variant(v)
        t = t_us(lambda: L.gemm_nt(A, B, C, bias=bias, R1=R1, R2=R2))
        t0 = t_us(lambda: L.gemm_nt(A, B, C, bias=bias))
        line += f'  variant {v}: {t:.1f} us ({(3 * M * N * 2 + M * 128) / t / 1e6:.2f} TB/s; no residuals {t0:.1f} us)'
    L.gemm_variant(2)
    print(line, flush=True)
