#!/usr/bin/env python
"""a4r_ln_fwd / a4r_ln_bwd at the ViT-B/16 step's shape (66 192 rows x 768, bf16; frozen parameters, residual gradient added) and at the text tower's
(40 448 rows, parameter gradients on): launch time and bytes moved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

def t_us(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device('cuda:0')
H = 768
for M, train in ((66192, False), (40448, True)):
    x = torch.randn(M, H, device=dev).bfloat16(); dy = torch.randn(M, H, device=dev).bfloat16(); dr = torch.randn(M, H, device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    g = torch.randn(H, device=dev); b = torch.randn(H, device=dev); st = torch.empty(2 * M, device=dev)
    dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
    tf = t_us(lambda: L.ln_fwd(x, g, b, 1e-12, y, st))
    if train: tb = t_us(lambda: L.ln_bwd(dy, x, st, g, dx, dgamma=dg, dbeta=db, dres=dr))
    else: tb = t_us(lambda: L.ln_bwd(dy, x, st, g, dx, dres=dr))
    by = M * H * 2
    print(f'M={M} parameter gradients={train}: ln_fwd {tf:.1f} us ({2 * by / tf / 1e6:.2f} TB/s)   ln_bwd {tb:.1f} us ({4 * by / tb / 1e6:.2f} TB/s)')
