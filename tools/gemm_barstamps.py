#!/usr/bin/env python
"""Per-wave arrival / release times at 16 consecutive barriers of the 256-tile GEMM's K loop (build: -DA4R_STAMP -DA4R_STAMP2).
usage: A4R_LIB_PATH=<lib> python tools/gemm_barstamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from adapter4rec_amd import _lib as L
dev = torch.device('cuda:0')
M, N, K = 40448, 768, 3072
L.gemm_variant(4)
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
C_ = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(30): L.gemm_nt(A, B, C_)
torch.cuda.synchronize()
n = 256 * 16 * 8 * 3
buf = (C.c_ulonglong * n)()
assert L.lib().a4r_debug_barstamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 8, 3).astype(np.int64)
ok = st[:, 0, 0, 0] > 0
st = st[ok]
print(f'{int(ok.sum())} workgroups with a second tile; cycles relative to the release of the previous barrier, median over workgroups')
rel = st[:, 1:, :, :] - st[:, :-1, :, 2].max(axis=2, keepdims=True)[..., None]      # vs the LAST wave's release of the previous barrier
for b in range(15):
    arr_w = np.median(rel[:, b, :, 0], axis=0)      # reached the waitcnt
    arr_b = np.median(rel[:, b, :, 1], axis=0)      # reached s_barrier
    out_b = np.median(rel[:, b, :, 2], axis=0)      # left s_barrier
    print(f'barrier {b + 65} (phase {(b + 1) % 4}): at waitcnt {arr_w.astype(int)}  at s_barrier {arr_b.astype(int)}  released {out_b.astype(int)}')
last = (st[:, 1:, :, 1].max(axis=2) - st[:, :-1, :, 2].max(axis=2))
first = (st[:, 1:, :, 1].min(axis=2) - st[:, :-1, :, 2].max(axis=2))
print('phase length (last release -> last arrival), median per barrier:', np.median(last, axis=0).astype(int))
print('first arrival:', np.median(first, axis=0).astype(int))
