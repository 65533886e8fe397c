#!/usr/bin/env python
"""Where does the 256-tile GEMM lose time?  Per-tile time for (a) the full problem, (b) A rows aliased to ONE 256-row panel
(every A byte an L2 hit: as_strided view, same kernel, same instruction stream), (c) a problem of one tile per CU."""
import sys, os
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
g = torch.Generator().manual_seed(1)
L.gemm_variant(2)
for N, K in ((768, 768), (768, 3072), (3072, 768)):
    for M in (40448, 65536, 256 * 85):
        A = torch.randn(M, K, generator=g).bfloat16().to(dev)
        B = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        tiles = (M // 256) * (N // 256)
        rounds = -(-tiles // 256)
        t = t_us(lambda: L.gemm_nt(A, B, C))
        print(f'N={N} K={K} M={M}: {t:7.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s  tiles={tiles} rounds={rounds} per-round {t/rounds:6.1f} us')
    M = 40448
    A1 = torch.randn(256, K, generator=g).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    g_ = L.GemmArgs()
    # hand-built descriptor: lda = 0 would alias rows, instead alias PANELS: M = 256 rows repeated via C offset loop is not
    # expressible, so use one launch per 256-row panel group?  Simplest: A panel of 256 rows, C of 256 rows, 158 tiles in N.
    Nw = 768 * 158 if N == 768 else N * 40
    Bw = (torch.randn(Nw, K, generator=g) * 0.05).bfloat16().to(dev)
    Cw = torch.zeros(256, Nw, dtype=torch.bfloat16, device=dev)
    tiles = Nw // 256
    t = t_us(lambda: L.gemm_nt(A1, Bw, Cw))
    print(f'  one A panel (L2-resident A), B streamed: N={Nw} K={K}: {t:7.1f} us {2.0*256*Nw*K/t/1e6:7.1f} TF/s tiles={tiles}')
