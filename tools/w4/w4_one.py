#!/usr/bin/env python
"""One NT GEMM shape, one kernel variant, n launches -- the program to put behind `rocprofv3 ... --` for counter passes.
usage: python3 tools/w4/w4_one.py <variant 8|9> <N> <K> [M=40448] [n=20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adapter4rec_amd import _lib as L

v, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 40448
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(3)
A = torch.randn(M, K, device=dev, generator=g).bfloat16()
B = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(N, device=dev)
L.gemm_variant(v)
for _ in range(n):
    L.gemm_nt(A, B, C, bias=bias)
torch.cuda.synchronize()
