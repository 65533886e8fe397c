#!/usr/bin/env python
"""a4r_attn_long_fwd at S = 197: launch time against the number of (item, head) workgroups (what one workgroup's lifetime is)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
S, nh, dh, H = 197, 12, 64, 768
for n_items in (11, 22, 43, 86, 171, 336):
    M = (n_items * S + 255) // 256 * 256
    qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
    out = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(n_items * nh * S, device=dev)
    f = lambda: L.attn_long_fwd(qkv, out, lse, n_items, S, nh, dh, 0, H, 2 * H, 1 / math.sqrt(dh))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f'items {n_items:4d}  workgroups {n_items * nh:5d} ({n_items * nh / 256:5.2f} per CU)  {us:7.1f} us')
