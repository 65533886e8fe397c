#!/usr/bin/env python
"""Time a4r_attn_long_fwd/bwd at the ViT-B/16 shape (336 items x 197 tokens x 12 heads, bf16) and the MAE shape (S=50)."""
import sys, os, math
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
for n_items, S in ((336, 197), (336, 50)):
    nh, dh, H = 12, 64, 768
    M = (n_items * S + 255) // 256 * 256
    qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
    out = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(M, H, device=dev).bfloat16()
    dqkv = torch.zeros_like(qkv)
    lse = torch.zeros(n_items * nh * S, device=dev); ws = torch.zeros_like(lse)
    sc = 1 / math.sqrt(dh)
    tf = t_us(lambda: L.attn_long_fwd(qkv, out, lse, n_items, S, nh, dh, 0, H, 2 * H, sc))
    tb = t_us(lambda: L.attn_long_bwd(qkv, out, dout, dqkv, lse, ws, n_items, S, nh, dh, 0, H, 2 * H, sc))
    fl = 4.0 * n_items * nh * S * S * dh
    by = 4.0 * n_items * S * H * 2
    print(f'S={S} items={n_items}: fwd {tf:.1f} us ({fl/tf/1e6:.1f} TF/s, {by/tf/1e6:.2f} TB/s)  bwd {tb:.1f} us ({2.5*fl/tb/1e6:.1f} TF/s, {2*by/tb/1e6:.2f} TB/s)')
