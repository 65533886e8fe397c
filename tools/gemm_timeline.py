#!/usr/bin/env python
"""Where a launch of the 256-tile GEMM spends its time (diagnostic build -DA4R_STAMP, `bash tools/gemm_stamps.sh` builds it): wave 0 of every
workgroup stamps s_memrealtime (100 MHz, one counter for the chip) at kernel entry, at the start / end of the K loop of its first three tiles,
after the last store of each of them was ISSUED and after the stores of its last tile have drained.  Printed: medians over the workgroups,
relative to the earliest entry stamp, and the event-timed duration of the launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(os.environ.get('M', 40448))
L.gemm_variant(4)
FORMS = os.environ.get('FORMS', 'plain').split(',')          # plain, bias, drop, res, gelu8, dmul8 (the step's epilogue forms)
for N, K, form in [(n, k, f) for f in FORMS for n, k in (((768, 3072), (768, 768), (3072, 768), (2304, 768)) if f == 'plain' else ((3072, 768),) if f in ('gelu8', 'dmul8') else ((768, 768), (768, 3072)))]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    if form == 'plain':
        run = lambda: L.gemm_nt(A, B, Cc)
    elif form == 'bias':
        run = lambda: L.gemm_nt(A, B, Cc, bias=bias)
    elif form == 'drop':
        run = lambda: L.gemm_nt(A, B, Cc, bias=bias, drop_p=0.1, drop_site=3, drop_seed=11)
    elif form == 'res':
        R1 = torch.randn(M, N, device=dev).bfloat16()
        run = lambda: L.gemm_nt(A, B, Cc, R1=R1)
    elif form == 'gelu8':
        C8 = torch.empty(M, N, device=dev, dtype=torch.uint8)
        run = lambda: L.gemm_nt(A, B, Cc, bias=bias, C2=C8, act=L.ACT_GELU, c2_deriv='q8')
    elif form == 'dmul8':
        P8 = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8)
        run = lambda: L.gemm_nt(A, B, Cc, Pre=P8, dact=L.DACT_MUL_Q8)
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = (C.c_ulonglong * (256 * 12))()
    assert L.lib().a4r_debug_timeline(buf) == 0
    tl = np.frombuffer(buf, dtype=np.uint64).reshape(256, 12).astype(np.int64)
    t0 = tl[:, 0].min()
    rel = (tl - t0) / 100.0                                    # us
    ntile = (M // 256) * (N // 256)
    two = np.array([b for b in range(256) if (ntile // 256 + (1 if b < ntile % 256 else 0)) >= 2 or True])
    def q(col, sel=slice(None)):
        v = rel[sel, col]
        return f'{np.median(v):6.1f} [{v.min():6.1f} {v.max():6.1f}]'
    print(f'{form} N={N} K={K}: launch {us:6.1f} us ({ntile} tiles = {ntile / 256:.2f} rounds)   us after the first entry: median [min max] over 256 workgroups')
    print(f'   entry {q(0)}')
    for t in range(min(3, (ntile + 255) // 256)):
        has = rel[:, 1 + 3 * t] > 0 if t else slice(None)
        d_k = tl[:, 2 + 3 * t] - tl[:, 1 + 3 * t]
        d_e = tl[:, 3 + 3 * t] - tl[:, 2 + 3 * t]
        sel = (tl[:, 2 + 3 * t] >= t0) & (tl[:, 1 + 3 * t] >= t0)
        print(f'   tile {t}: K loop starts {q(1 + 3 * t, sel)}  ends {q(2 + 3 * t, sel)}  stores issued {q(3 + 3 * t, sel)}   '
              f'K loop {np.median(d_k[sel]) / 100:5.1f} us, epilogue issue {np.median(d_e[sel]) / 100:5.1f} us ({int(sel.sum())} workgroups)')
    sel = tl[:, 10] >= t0
    print(f'   last stores drained {q(10, sel)}   last workgroup done at {rel[sel, 10].max():6.1f} us of {us:6.1f}')
