#!/usr/bin/env python
"""What the partial last round of a launch costs: the plain bf16 GEMM at M = 65 536 (256 row panels: whole rounds on 256 CUs) against the image
tower's M = 66 304 (259 panels) for its four (N, K) shapes; A4R_GEMM_TAIL selects the largest short-tile height (0 = a whole extra round).

CAVEAT (round 4): the launches are timed back to back and do not depend on each other, so the thin last round of one overlaps the first round of the next
and "no tail" looks 5 - 10 % cheaper here than it is in the training step, where the consumer of a GEMM waits for its last tile: with the short-tile tail
switched off for partial rounds below a quarter (the policy this table suggests) the ViT-B/16 + LoRA step ran 31.40 against 31.35 ms in bf16 and 28.0
against 27.3 ms with --dtype fp8, same box, three alternating runs -- the tail stays as it is."""
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
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    out = []
    for M in (65536, 66304, 40448):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        t = t_us(lambda: L.gemm_nt(A, B, C))
        out.append(f'M={M}: {t:7.1f} us ({t / M * 1e3:.3f} ns/row, {2.0 * M * N * K / t / 1e6:6.0f} TF/s)')
    print(f'N={N} K={K}   ' + '   '.join(out))
