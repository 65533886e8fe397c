#!/usr/bin/env python
"""a4r_colsum on the step's shapes (bias gradients): us and TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
for M, N in ((66304, 768), (40448, 768), (40448, 64)):
    X = torch.randn(M, N, device=dev).bfloat16()
    out = torch.zeros(N, device=dev)
    for _ in range(5):
        L.colsum(X, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        L.colsum(X, out)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    print(f'colsum [{M}, {N}] bf16: {us:6.1f} us  {M * N * 2 / us / 1e6:5.2f} TB/s')
