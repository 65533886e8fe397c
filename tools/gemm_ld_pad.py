#!/usr/bin/env python
"""Does the row stride of the operands matter to the 256-tile GEMM?  (L2 / fabric channel mapping: a K-tile is one 128-byte line per row, rows
lda * 2 bytes apart -- with K = 3072 that is 48 lines, K = 768: 12 lines.)  Same product, operands as views of wider buffers.
usage: python tools/gemm_ld_pad.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
M = 40448


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def view(rows, cols, pad, scale=1.0):
    buf = (torch.randn(rows, cols + pad, generator=g) * scale).bfloat16().to(dev)
    return buf[:, :cols]


for N, K in ((768, 3072), (768, 2304), (3072, 768), (2304, 768), (768, 768)):
    res = []
    for pa, pb, pc in ((0, 0, 0), (64, 0, 0), (128, 0, 0), (192, 0, 0), (256, 0, 0), (320, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (0, 0, 64), (0, 0, 0)):
        A, B, Cc = view(M, K, pa), view(N, K, pb, 0.05), view(M, N, pc)
        t = timed(lambda: L.gemm_nt(A, B, Cc))
        tv = timed(lambda: torch.matmul(A, B.t(), out=Cc)) if pc == 0 else float('nan')
        res.append(f'pad A/B/C {pa:3d}/{pb:3d}/{pc:3d}: a4r {t:6.1f} us  vendor {tv:6.1f} us')
        del A, B, Cc
    print(f'M={M} N={N} K={K}')
    for r in res:
        print('   ', r)
