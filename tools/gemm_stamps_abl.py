#!/usr/bin/env python
"""K-loop cycles per K-tile of the 256-tile GEMM for ablation builds (-DA4R_STAMP -DA4R_ABL=<bits>: 1 no LDS-DMA, 4 no ds_read, 8 no barrier).
usage: A4R_LIB_PATH=<lib> python tools/gemm_stamps_abl.py"""
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
buf = (C.c_ulonglong * (1024 * 8))()
assert L.lib().a4r_debug_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:256].astype(np.int64)
two = st[:, 6] > st[:, 0]
kl = np.median(st[two, 5] - st[two, 4])
print(f'{os.path.basename(os.environ.get("A4R_LIB_PATH", "default"))}: K-loop#2 {int(kl)} cycles = {kl / (K // 64):.0f} per K-tile (2048 MFMA cycles)')
