#!/usr/bin/env python
"""In-kernel stamps of the four-wave GEMM's K loop (a library built with -DA4R_W4_STAMP: bash tools/w4/w4_variants.sh <variant> with STAMP=1):
shader cycles per K-tile (2 048 = the matrix pipe's own time for 128 MFMAs 16x16x32), the clock the chip holds inside the loop and the
epilogue's time, medians over workgroups.  usage: A4R_LIB_PATH=tools/_ab/liba4r_w4_<v>_st.so python tools/w4/w4_stamps.py [M=40448]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
g = torch.Generator(device=dev).manual_seed(3)
for N, K in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    L.gemm_variant(9)
    for _ in range(200):                      # ~ tens of ms of back-to-back launches: the clock settles
        L.gemm_nt(A, B, Cc, bias=bias)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 6, dtype=np.uint64)
    rc = L.lib().a4r_debug_w4_stamps(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0
    s = buf.reshape(256, 2, 3, 2).astype(np.int64)
    nk = K // 64
    for t in range(2):
        ok = s[:, t, 0, 0] > 0
        cyc = (s[ok, t, 1, 0] - s[ok, t, 0, 0])
        rt = (s[ok, t, 1, 1] - s[ok, t, 0, 1])            # 100 MHz ticks
        ecyc = (s[ok, t, 2, 0] - s[ok, t, 1, 0])
        ert = (s[ok, t, 2, 1] - s[ok, t, 1, 1])
        if ok.sum() == 0:
            continue
        print(f'N={N:4d} K={K:4d} tile {t}: {int(ok.sum()):3d} wgs  K loop {np.median(rt) / 100:7.2f} us = {np.median(cyc) / nk:7.0f} cycles per K-tile (min {cyc.min() / nk:6.0f}, max {cyc.max() / nk:6.0f}) '
              f'at {np.median(cyc / np.maximum(rt, 1)) * 100 / 1e3:5.2f} GHz | epilogue {np.median(ert) / 100:6.2f} us = {np.median(ecyc):7.0f} cycles')
