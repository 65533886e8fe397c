#!/usr/bin/env python
"""Where does a 256-tile workgroup spend its time?  Needs a diagnostic build of a4r_gemm256.hip with -DA4R_STAMP
(make FLAGS_a4r_gemm256=-DA4R_STAMP); prints median cycle counts between the stamps of the first two tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from adapter4rec_amd import _lib as L
dev = torch.device('cuda:0')
M = 40448
L.gemm_variant(4)
for N, K, kw in ((768, 768, {}), (768, 3072, {}), (768, 64, {}), (3072, 768, dict(act=2, c2=True)), (768, 768, dict(r1=True, drop=True))):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    C_ = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev)
    C2 = torch.empty_like(C_) if kw.get('c2') else None
    R1 = torch.randn(M, N, device=dev).bfloat16() if kw.get('r1') else None
    def run():
        L.gemm_nt(A, B, C_, bias=bias, C2=C2, act=kw.get('act', 0), c2_deriv=bool(kw.get('c2')), R1=R1,
                  drop_p=0.1 if kw.get('drop') else 0.0, drop_site=1, drop_seed=5, drop_first=True)
    for _ in range(20): run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8))()
    assert L.lib().a4r_debug_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:256].astype(np.int64)
    two = st[:, 6] > st[:, 0]                       # workgroups that ran two tiles
    d = lambda a, b: int(np.median(st[two, b] - st[two, a]))
    print(f'N={N} K={K} {kw}: K-loop#1 (incl. first prologue) {d(0,1)} | epilogue issue {d(1,2)} | vmcnt(0) drain {d(2,3)} | barrier {d(3,4)} | '
          f'K-loop#2 {d(4,5)} | epilogue#2 issue {d(5,6)}  [cycles, median over {int(two.sum())} workgroups; 100 MHz ticks if s_memtime is the constant clock]')
