#!/usr/bin/env python
"""a4r_lora_bwd_fused against the five products it replaces at the image tower's rows (66 304 x 768, bf16, r = 8 + 8) and the text tower's (40 448)."""
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
H, rp, T = 768, 64, torch.bfloat16
for M in (66304, 40448):
    x = torch.randn(M, H, device=dev).to(T); dqkv = (torch.randn(M, 3 * H, device=dev) * 0.1).to(T)
    dqa, dqb = dqkv[:, :H], dqkv[:, 2 * H:]
    A, BTa, BTb = (torch.zeros(rp, H, dtype=T, device=dev) for _ in range(3))
    A[0:8] = (torch.randn(8, H, device=dev) * 0.05).to(T); A[16:24] = (torch.randn(8, H, device=dev) * 0.05).to(T)
    BTa[0:8] = (torch.randn(8, H, device=dev) * 0.05).to(T); BTb[16:24] = (torch.randn(8, H, device=dev) * 0.05).to(T)
    sA = torch.zeros(rp, H, device=dev); sBa = torch.zeros(H, rp, device=dev); sBb = torch.zeros(H, rp, device=dev)
    ones = torch.zeros(rp, device=dev); ones[32] = 1
    t = torch.zeros(M, rp, dtype=T, device=dev); dt = torch.zeros(M, rp, dtype=T, device=dev)
    def five():
        L.gemm_nt(x, A, t, bias=ones, M=M)
        L.gemm_nt(dqa, BTa, dt, alpha=0.125, M=M)
        L.gemm_nt(dqb, BTb, dt, alpha=0.125, R1=dt, M=M)
        L.gemm_tn2(dqa, t, sBa, dqb, t, sBb, M=M)
        L.gemm_tn(dt, x, sA, M=M)
    def fused():
        L.lora_bwd_fused(x, dqa, dqb, A[0:8], A[16:24], BTa[0:8], BTb[16:24], 0.125, 0.125, sA[0:8], sA[16:24], sBa[:, 0:8], sBb[:, 16:24], sBa[:, 32], sBb[:, 32], M)
    a, b = t_us(five), t_us(fused)
    by = 3.0 * M * H * 2
    print(f'M={M}: five launches {a:7.1f} us   fused {b:7.1f} us ({by / b / 1e6:.2f} TB/s on its {by / 1e6:.0f} MB)')
