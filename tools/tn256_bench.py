#!/usr/bin/env python
"""a4r_gemm_tn / a4r_gemm_tn_bias on the weight-gradient shapes of TRAINABLE backbone Linears (--fine_tune_to all / Pretraining): us and TF/s per
product at M token rows.  A4R_TN256=0 python tools/tn256_bench.py = the 64-tile kernel on the same shapes; A4R_TN256_WGS=n: workgroups per launch.
usage: python tools/tn256_bench.py [M=40448] [iters=30]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
it = int(sys.argv[2]) if len(sys.argv) > 2 else 30


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


print(f'A4R_TN256={os.environ.get("A4R_TN256", "1")} A4R_TN256_WGS={os.environ.get("A4R_TN256_WGS", "256")}')
for P, Q in ((768, 768), (3072, 768), (768, 3072), (2304, 768)):
    X = (torch.randn(M, P, device=dev) * 0.05).bfloat16()
    Y = torch.randn(M, Q, device=dev).bfloat16()
    out, xs = torch.zeros(P, Q, device=dev), torch.zeros(P, device=dev)
    t0 = timed(lambda: L.gemm_tn(X, Y, out))
    t1 = timed(lambda: L.gemm_tn_bias(X, Y, out, xs))
    tv = timed(lambda: torch.matmul(X.t(), Y))
    fl = 2.0 * M * P * Q
    print(f'dW [{P} x {Q}] over M={M}: gemm_tn {t0:7.1f} us {fl / t0 / 1e6:7.1f} TF/s | + bias sums {t1:7.1f} us | vendor (bf16 out) {tv:7.1f} us {fl / tv / 1e6:7.1f} TF/s')
