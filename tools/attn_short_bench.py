#!/usr/bin/env python
"""Time a4r_attn_fwd / a4r_attn_bwd at the text tower's shape (1344 items x 30 tokens x 12 heads x 64, bf16, dropout on)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L


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


dev = torch.device('cuda:0')
n_items, S, nh, dh, H = 1344, 30, 12, 64, 768
M = (n_items * S + 255) // 256 * 256
qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
out = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
dout = torch.randn(M, H, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv)
mask = torch.ones(n_items, S, device=dev)
sc = 1 / math.sqrt(dh)
neg = float(torch.finfo(torch.float32).min)
tf = t_us(lambda: L.attn_fwd(qkv, out, mask, n_items, S, nh, dh, 0, H, 2 * H, False, sc, neg, drop_p=0.1, drop_site=1, drop_seed=5))
tb = t_us(lambda: L.attn_bwd(qkv, dout, dqkv, mask, n_items, S, nh, dh, 0, H, 2 * H, False, sc, neg, drop_p=0.1, drop_site=1, drop_seed=5))
rows = n_items * S
bf, bb = rows * H * 2 * 4, rows * H * 2 * 7          # fwd: q, k, v in + ctx out; bwd: q, k, v, dO in + dq, dk, dv out
print(f'attn_fwd {tf:6.1f} us  {bf / tf / 1e6:5.2f} TB/s ({bf / 1e6:.0f} MB)   attn_bwd {tb:6.1f} us  {bb / tb / 1e6:5.2f} TB/s ({bb / 1e6:.0f} MB)')
