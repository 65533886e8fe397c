#!/usr/bin/env python
"""Time of ONE short tile per CU against one full 256 x 256 tile per CU: launches that are all short tiles (N = 1024: 4 column panels x 64 short
tiles of 32 kp rows = 256 workgroups) at kp = 1..7; run once with A4R_GEMM_TAIL=0 (the same rows as 8 kp full tiles on 32 kp CUs) and once with
the default.  usage: python tools/gemm_tail_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
t = torch.bfloat16


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


L.gemm_variant(4)
if os.environ.get('A4R_GEMM_TAIL') is None:
    L.gemm_tail_max(7)          # every height (the default stops at kp = 3)
g = torch.Generator(device=dev).manual_seed(3)
for K in (768, 3072):
    for kp in (1, 2, 3, 4, 5, 6, 7, 8):
        M, N = 2048 * kp, 1024
        A = torch.randn(M, K, device=dev, generator=g).to(t)
        B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(t)
        C = torch.empty(M, N, device=dev, dtype=t)
        us = t_us(lambda: L.gemm_nt(A, B, C))
        print(f'K={K} M={M} plan={L.gemm_tail_plan(M, N)} {us:7.1f} us')
