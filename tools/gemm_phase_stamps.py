#!/usr/bin/env python
"""Where a K-tile of the 256-tile GEMM spends its time, per phase and ping-pong half (diagnostic build -DA4R_PHASE_STAMP; see a4r_gemm256.hip).
Build + run on the GPU box:   bash tools/gemm_phase_stamps.sh
usage: A4R_LIB_PATH=tools/_ab/liba4r_pst.so python tools/gemm_phase_stamps.py [M N K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (40448, 768, 3072)
t = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).to(t)
B = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(t)
Cc = torch.empty(M, N, device=dev, dtype=t)
for _ in range(200):                                   # > 30 ms of back-to-back launches: the clock the chip holds under this load
    L.gemm_nt(A, B, Cc)
torch.cuda.synchronize()
buf = np.zeros(64 * 2 * 8 * 8, dtype=np.uint64)
lib = L.lib()
lib.a4r_debug_phase_stamps.argtypes = [C.c_void_p]
assert lib.a4r_debug_phase_stamps(buf.ctypes.data) == 0
s = buf.reshape(64, 2, 8, 8).astype(np.int64)
names = ['issue reads+DMA', 'vmcnt wait', 'barrier A', 'lgkmcnt wait', 'MFMA segment', 'barrier B']
print(f'M={M} N={N} K={K}: median shader cycles over 64 workgroups; phases of K-tile 4 (ring buffer 0) and 5 (buffer 1)')
print('half phase | ' + ' | '.join(f'{n:>15s}' for n in names) + ' |  phase total | to next phase')
for h in range(2):
    for p in range(8):
        d = np.diff(s[:, h, p, :7], axis=1)
        med = np.median(d, axis=0)
        tot = np.median(s[:, h, p, 6] - s[:, h, p, 0])
        nxt = np.median(s[:, h, p + 1, 0] - s[:, h, p, 6]) if p < 7 else float('nan')
        print(f'   {h}   {p % 4} ({p // 4}) | ' + ' | '.join(f'{v:15.0f}' for v in med) + f' | {tot:12.0f} | {nxt:8.0f}')
for h in range(2):
    kt = np.median(s[:, h, 4, 0] - s[:, h, 0, 0])
    print(f'half {h}: K-tile period {kt:.0f} cycles (2 048 = the MFMA pipe busy all the time)')
