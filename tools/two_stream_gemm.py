#!/usr/bin/env python
"""Would two half-batches on two streams hide the 256-tile GEMMs' partial last rounds?  One encoder layer's eight large GEMMs (plain
epilogues) back to back at M rows on one stream, against the same chain at M/2 rows on each of two streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
t = torch.bfloat16
SHAPES = ((2304, 768), (768, 768), (3072, 768), (768, 3072), (3072, 768), (768, 3072), (768, 768), (768, 2304))


def chain(M):
    bufs = []
    for N, K in SHAPES:
        bufs.append((torch.randn(M, K, device=dev).to(t), (torch.randn(N, K, device=dev) * 0.05).to(t), torch.empty(M, N, device=dev, dtype=t)))
    def run():
        for A, B, C in bufs:
            L.gemm_nt(A, B, C)
    return run


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
one = chain(M)
print(f'one stream, M = {M}: {timed(one):8.1f} us per layer')
h1, h2 = chain(M // 2), chain(M // 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        h1()
    with torch.cuda.stream(s2):
        h2()
    cur.wait_stream(s1); cur.wait_stream(s2)


print(f'two streams, M / 2 each: {timed(two):8.1f} us per layer')
seq = lambda: (h1(), h2())
print(f'one stream, two half chains in sequence: {timed(seq):8.1f} us per layer')
