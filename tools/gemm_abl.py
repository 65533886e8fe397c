#!/usr/bin/env python
"""Ablation of the 256-tile GEMM main loop: libraries built with -DA4R_ABL=<bits> (1 no LDS-DMA, 2 no MFMA, 4 no ds_read).
usage: python tools/gemm_abl.py <lib.so>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
M = 256 * 85
for N, K in ((768, 768), (768, 3072)):
    A = torch.randn(M, K, generator=g).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3): L.gemm_nt(A, B, C)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): L.gemm_nt(A, B, C)
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    print(f'{os.path.basename(L.LIB_PATH)} N={N} K={K} one tile per CU: {t:7.1f} us  ({t / (K // 64):.2f} us per K-tile incl. prologue/epilogue)')
