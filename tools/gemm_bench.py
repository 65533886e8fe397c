#!/usr/bin/env python
"""Micro-benchmark of a4r_gemm_nt on the BERT-base shapes of the training step (GPU box only).
Interleaves the staging variants in ONE process (guide rule 24) and prints TFLOP/s per shape."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

def main():
    dev = torch.device('cuda:0')
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304), (768, 64), (64, 768)]
    g = torch.Generator().manual_seed(1)
    res = {}
    for N, K in shapes:
        A = (torch.randn(M, K, generator=g) * 1.0).bfloat16().to(dev)
        B = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        bias = torch.zeros(N, device=dev)
        for rnd in range(3):
            for v in ([int(x) for x in os.environ['A4R_VARIANTS'].split(',')] if 'A4R_VARIANTS' in os.environ else (2, 3)):
                L.gemm_variant(v)
                for _ in range(3):
                    L.gemm_nt(A, B, C, bias=bias)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.gemm_nt(A, B, C, bias=bias)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 10 * 1e-3
                res.setdefault((N, K, v), []).append(2.0 * M * N * K / t / 1e12)
    L.gemm_variant(2)      # the default: variant 3 (four 128 x 128 waves, a4r_gemm256w4.hip) measured 20-40 % slower
    for (N, K, v), tf in sorted(res.items()):
        print(f'M={M} N={N:5d} K={K:5d} variant={v}: median {sorted(tf)[len(tf)//2]:8.1f} TF/s  (min {min(tf):.1f} max {max(tf):.1f})')

if __name__ == '__main__':
    main()
