#!/usr/bin/env python
"""One GEMM shape, N launches (for rocprofv3 --pmc runs on the GPU box): python tools/gemm_one.py M N K variant iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L
M, N, K, v, it = [int(x) for x in sys.argv[1:6]]
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).bfloat16().to(dev)
B = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
L.gemm_variant(v)
for _ in range(it):
    L.gemm_nt(A, B, C)
torch.cuda.synchronize()
