#!/usr/bin/env python
"""In-kernel stamps of the one-pass long-attention backward (library built with -DA4R_OP_STAMP: TAG=st EXTRA=-DA4R_OP_STAMP=1 bash tools/_ab/op_abl.sh):
shader cycles between the six points of wave 0, medians over the workgroups' third pair.  usage: A4R_LIB_PATH=tools/_ab/liba4r_op_st.so python tools/op_stamps.py"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
n_items, S, nh, dh, H = 336, 197, 12, 64, 768
M = (n_items * S + 255) // 256 * 256
qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
out = torch.randn(M, H, device=dev).bfloat16()
dout = torch.randn(M, H, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv)
lse = torch.zeros(n_items * nh * S, device=dev)
ws = torch.zeros_like(lse)
for _ in range(30):
    L.attn_long_bwd(qkv, out, dout, dqkv, lse, ws, n_items, S, nh, dh, 0, H, 2 * H, 1 / math.sqrt(dh))
torch.cuda.synchronize()
buf = np.zeros(256 * 8, dtype=np.uint64)
assert L.lib().a4r_debug_op_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.reshape(256, 8).astype(np.int64)
names = ['wait for DMA / K / V / O + barrier B0', 'delta, stats, K image (B1, B2)', 'DMA issue + 7 steps', 'requests + final barrier', 'dk / dv stores']
order = [0, 1, 2, 3, 5, 4]
for i, n in enumerate(names):
    d = s[:, order[i + 1]] - s[:, order[i]]
    print(f'{n:32s} median {np.median(d):8.0f} cycles  (min {d.min():7d}, max {d.max():7d})')
print(f'{"  of which: issuing the requests":32s} median {np.median(s[:, 6] - s[:, 3]):8.0f} cycles; consumer wave reaches the final barrier {np.median(s[:, 7] - s[:, 3]):8.0f} cycles after wave 0 left its steps')
tot = s[:, 4] - s[:, 0]
print(f'{"whole workgroup":32s} median {np.median(tot):8.0f} cycles')
