#!/usr/bin/env python
"""a4r_gemm_tn on the adapter weight-gradient shapes of the step: dW_up [768, 64] = dv^T z, dW_down [64, 768] = dzp^T h; us, TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
for P, Q in ((768, 64), (64, 768)):
    X = torch.randn(M, P, device=dev).bfloat16()
    Y = torch.randn(M, Q, device=dev).bfloat16()
    out = torch.zeros(P, Q, device=dev)
    for _ in range(5):
        L.gemm_tn(X, Y, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        L.gemm_tn(X, Y, out)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    print(f'gemm_tn M={M} [{P} x {Q}]: {us:6.1f} us  {M * (P + Q) * 2 / us / 1e6:5.2f} TB/s')

X1, Y1 = torch.randn(M, 768, device=dev).bfloat16(), torch.randn(M, 64, device=dev).bfloat16()
X2, Y2 = torch.randn(M, 64, device=dev).bfloat16(), torch.randn(M, 768, device=dev).bfloat16()
C1, C2 = torch.zeros(768, 64, device=dev), torch.zeros(64, 768, device=dev)
for _ in range(5):
    L.gemm_tn2(X1, Y1, C1, X2, Y2, C2)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    L.gemm_tn2(X1, Y1, C1, X2, Y2, C2)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / 30 * 1e3
print(f'gemm_tn2 M={M} both products in one launch: {us:6.1f} us  {M * (768 + 64) * 2 * 2 / us / 1e6:5.2f} TB/s')
# + the bias sums from the same pass (what the step launches since the end of round 3)
s1, s2 = torch.zeros(768, device=dev), torch.zeros(64, device=dev)
for _ in range(5):
    L.gemm_tn2(X1, Y1, C1, X2, Y2, C2, xsum1=s1, xsum2=s2)
torch.cuda.synchronize()
a.record()
for _ in range(30):
    L.gemm_tn2(X1, Y1, C1, X2, Y2, C2, xsum1=s1, xsum2=s2)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / 30 * 1e3
print(f'gemm_tn2 + column sums M={M}: {us:6.1f} us  {M * (768 + 64) * 2 * 2 / us / 1e6:5.2f} TB/s   (A4R_TN2_WGS={os.environ.get("A4R_TN2_WGS", "384")})')
