#!/usr/bin/env python
"""Schedule sweep of the four-wave stream GEMM (a4r_gemm256s.hip built with -DA4R_SCHED_SWEEP): a4r_gemm_sched(k) selects
(F1 read spacing, WAR barrier, first DMA slot, slot spacing, RAW barrier).  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
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
M = 40448
scheds = [int(v) for v in os.environ.get('SCHEDS', '0,1,2,3,4').split(',')]
for N, K in ((768, 3072), (768, 768), (3072, 768), (768, 64)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    C_ = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev)
    R1 = torch.randn(M, N, device=dev).bfloat16()
    f = 2.0 * M * N * K
    best, bestd, ok = {}, {}, {}
    L.gemm_variant(4)
    L.gemm_nt(A, B, C_, bias=bias); ref = C_.clone()
    for rnd in range(3):                                   # interleaved rounds, minimum: clocks drift by several % within a run
        for k in [-1] + scheds:
            L.gemm_variant(4 if k < 0 else 5)
            if k >= 0: L.lib().a4r_gemm_sched(C.c_int(k))
            t = t_us(lambda: L.gemm_nt(A, B, C_, bias=bias))
            ok[k] = torch.equal(ref, C_)
            td = t_us(lambda: L.gemm_nt(A, B, C_, bias=bias, R1=R1, drop_p=0.1, drop_site=1, drop_seed=5, drop_first=True))
            best[k] = min(best.get(k, 1e9), t); bestd[k] = min(bestd.get(k, 1e9), td)
    L.lib().a4r_gemm_sched(C.c_int(0)); L.gemm_variant(2)
    print(f'N={N} K={K}: ' + ' | '.join(f'{"v4" if k < 0 else "s%d" % k} {best[k]:.1f} us ({f/best[k]/1e6:.0f} TF/s; +R1+drop {bestd[k]:.1f}){"" if ok[k] else " DIFF"}' for k in [-1] + scheds), flush=True)
