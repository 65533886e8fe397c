#!/usr/bin/env python
"""Shader cycles per K loop of the 256-tile GEMM and the clock the chip holds while it runs (diagnostic build -DA4R_STAMP, see
tools/gemm_stamps.sh; optional -DA4R_ABL=n ablations): wave 0 of every workgroup stamps (s_memtime, s_memrealtime) around the K loop
of its first output tiles.  clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = 40448
L.gemm_variant(4)
for N, K in ((768, 3072), (768, 768), (3072, 768)):
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(200):                      # ~2 s of back-to-back launches: the clock has settled
        L.gemm_nt(A, B, Cc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.gemm_nt(A, B, Cc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = (C.c_ulonglong * (256 * 4 * 4))()
    assert L.lib().a4r_debug_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 4).astype(np.int64)[:, 0]      # first tile of every workgroup
    cyc = st[:, 2] - st[:, 0]
    rt = st[:, 3] - st[:, 1]
    nk = K // 64
    ghz = np.median(cyc / np.maximum(rt, 1)) * 0.1
    print(f'N={N} K={K}: launch {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s | K loop {int(np.median(cyc))} cycles = {np.median(cyc) / nk:7.1f} per K-tile '
          f'(2048 = MFMA-bound) | {np.median(rt) / 100:6.2f} us | clock {ghz:.2f} GHz')
