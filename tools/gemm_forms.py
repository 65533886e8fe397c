#!/usr/bin/env python
"""a4r_gemm_nt on the training step's shapes WITH the epilogue forms the step uses (plain + bias, residual + dropout,
GELU + derivative output, * saved derivative), us and TF/s per launch; `vendor` = torch.nn.functional.linear (hipBLASLt) on the
plain form for orientation (measurement only: the product never calls it).  A4R_GEMM_BAND=0 / n selects the tile map.
usage: python tools/gemm_forms.py [M=40448]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
PLAIN = len(sys.argv) > 2          # ablation runs: plain epilogue on three shapes, no vendor column
t = torch.bfloat16


def t_us(fn, n=20):
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


g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(t)
rows = []
CASES = (('qkv', 2304, 768, 'plain'), ('attn-out', 768, 768, 'drop'), ('ffn-up', 3072, 768, 'gelu'), ('ffn-down', 768, 3072, 'drop'),
                         ('ffn-up q8', 3072, 768, 'gelu8'), ('ffn-up q8 t', 3072, 768, 'gelu8t'), ('d ffn-down', 3072, 768, 'dmul'), ('d ffn-dn q8', 3072, 768, 'dmul8'), ('d ffn-dn q8t', 3072, 768, 'dmul8t'), ('d ffn-up', 768, 3072, 'res'), ('d attn-out', 768, 768, 'plain'), ('d qkv', 768, 2304, 'res'))
if PLAIN:
    CASES = (('k768', 768, 768, 'plain'), ('k3072', 768, 3072, 'plain'), ('n3072', 3072, 768, 'plain'))
for name, N, K, form in CASES:
    A, B = R(M, K), R(N, K, sc=0.05)
    C, C2, R1, Pre = (torch.empty(M, N, device=dev, dtype=t) for _ in range(4))
    R1.normal_(); Pre.normal_()
    bias = torch.zeros(N, device=dev)
    if form == 'plain':
        f = lambda: L.gemm_nt(A, B, C, bias=bias)
    elif form == 'drop':
        f = lambda: L.gemm_nt(A, B, C, bias=bias, drop_p=0.1, drop_site=3, drop_seed=11)
    elif form == 'gelu':
        f = lambda: L.gemm_nt(A, B, C, bias=bias, C2=C2, act=L.ACT_GELU, c2_deriv=True)
    elif form in ('gelu8', 'gelu8t'):          # 't': the 8-bit derivative in the 256-tile kernel's own order (what the engine uses since round 3)
        C8 = torch.empty(M, N, device=dev, dtype=torch.uint8)
        f = lambda: L.gemm_nt(A, B, C, bias=bias, C2=C8, act=L.ACT_GELU, c2_deriv='q8', q8_tiled=form.endswith('t'))
    elif form in ('dmul8', 'dmul8t'):
        P8 = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8)
        f = lambda: L.gemm_nt(A, B, C, Pre=P8, dact=L.DACT_MUL_Q8, q8_tiled=form.endswith('t'))
    elif form == 'dmul':
        f = lambda: L.gemm_nt(A, B, C, Pre=Pre, dact=L.DACT_MUL)
    else:
        f = lambda: L.gemm_nt(A, B, C, R1=R1)
    ta = t_us(f)
    tp = t_us(lambda: L.gemm_nt(A, B, C, bias=bias)) if form != 'plain' else ta      # the same product with the plain epilogue
    bb = bias.to(t)
    tv = t_us(lambda: torch.nn.functional.linear(A, B, bb)) if not PLAIN else float('nan')
    fl = 2.0 * M * N * K
    print(f'{name:11s} M={M} N={N:4d} K={K:4d} {form:6s}: a4r {ta:7.1f} us {fl / ta / 1e6:7.1f} TF/s (plain {tp:7.1f} us) | vendor plain {tv:7.1f} us {fl / tv / 1e6:7.1f} TF/s')
