#!/usr/bin/env python
"""The four-wave hand-scheduled 256-tile GEMM (a4r_gemm_variant 9, a4r_gemm256w4.hip) against the eight-wave kernel (variant 8) on the
training step's shapes and epilogue forms: outputs must be BIT-EQUAL (same K order per element, same epilogue text), then us per launch of
both, interleaved in one process (and the vendor library on the plain form for orientation, measurement only).
usage: python tools/w4/w4_check.py [M=40448] [rounds=3] [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adapter4rec_amd import _lib as L

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
QUICK = len(sys.argv) > 3
t = torch.bfloat16


def t_us(fn, n=20):
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


g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(t)
CASES = (('qkv', 2304, 768, 'plain'), ('attn-out', 768, 768, 'drop'), ('ffn-up q8 t', 3072, 768, 'gelu8t'), ('ffn-down', 768, 3072, 'drop'),
         ('d ffn-dn q8t', 3072, 768, 'dmul8t'), ('d ffn-up', 768, 3072, 'res'), ('d attn-out', 768, 768, 'plain'), ('d qkv', 768, 2304, 'res'),
         ('ffn-up', 3072, 768, 'gelu'), ('d ffn-down', 3072, 768, 'dmul'), ('k3072 plain', 768, 3072, 'plain'), ('n3072 plain', 3072, 768, 'plain'))
if QUICK:
    CASES = (('k768', 768, 768, 'plain'), ('k3072', 768, 3072, 'plain'))
ok_all = True
for name, N, K, form in CASES:
    A, B = R(M, K), R(N, K, sc=0.05)
    R1, Pre = R(M, N), R(M, N)
    bias = torch.randn(N, device=dev, generator=g)
    P8 = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8, generator=g)

    def make(C, C2, C8):
        if form == 'plain':
            return lambda: L.gemm_nt(A, B, C, bias=bias)
        if form == 'drop':
            return lambda: L.gemm_nt(A, B, C, bias=bias, R1=R1, drop_p=0.1, drop_site=3, drop_seed=11)
        if form == 'gelu':
            return lambda: L.gemm_nt(A, B, C, bias=bias, C2=C2, act=L.ACT_GELU, c2_deriv=True)
        if form == 'gelu8t':
            return lambda: L.gemm_nt(A, B, C, bias=bias, C2=C8, act=L.ACT_GELU, c2_deriv='q8', q8_tiled=True)
        if form == 'dmul8t':
            return lambda: L.gemm_nt(A, B, C, Pre=P8, dact=L.DACT_MUL_Q8, q8_tiled=True)
        if form == 'dmul':
            return lambda: L.gemm_nt(A, B, C, Pre=Pre, dact=L.DACT_MUL)
        return lambda: L.gemm_nt(A, B, C, R1=R1)

    outs = []
    fns = []
    for v in (8, 9):
        C = torch.full((M, N), float('nan'), device=dev, dtype=t)
        C2 = torch.full((M, N), float('nan'), device=dev, dtype=t)
        C8 = torch.zeros(M, N, device=dev, dtype=torch.uint8)
        f = make(C, C2, C8)
        L.gemm_variant(v)
        f()
        torch.cuda.synchronize()
        outs.append((C, C2, C8))
        fns.append(f)
    L.gemm_variant(8)
    eqC = torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    eq2 = torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16)) if form == 'gelu' else True
    eq8 = torch.equal(outs[0][2], outs[1][2]) if form == 'gelu8t' else True
    ok = eqC and eq2 and eq8
    ok_all &= ok
    if not ok:
        d = (outs[0][0].float() - outs[1][0].float())
        bad = (outs[0][0].view(torch.int16) != outs[1][0].view(torch.int16))
        nb = int(bad.sum())
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(f'   MISMATCH C: {nb} elements, max |d| {float(d.nan_to_num(1e9).abs().max()):.4g}, rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({rows.numel()}), cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({cols.numel()}); nan in w4: {int(outs[1][0].isnan().sum())}')
    ts = [[], []]
    for _ in range(ROUNDS):
        for i, v in enumerate((8, 9)):
            L.gemm_variant(v)
            ts[i].append(t_us(fns[i]))
    L.gemm_variant(8)
    bb = bias.to(t)
    tv = t_us(lambda: torch.nn.functional.linear(A, B, bb))
    fl = 2.0 * M * N * K
    m8, m9 = min(ts[0]), min(ts[1])
    print(f'{name:13s} N={N:4d} K={K:4d} {form:7s} {"bit-equal" if ok else "DIFFERENT"}: w8 {m8:7.1f} us {fl / m8 / 1e6:6.0f} TF/s | w4 {m9:7.1f} us {fl / m9 / 1e6:6.0f} TF/s ({m9 / m8:5.3f}x) | vendor plain {tv:7.1f} us {fl / tv / 1e6:6.0f} TF/s'
          f'   [w8 {" ".join("%.1f" % x for x in ts[0])} | w4 {" ".join("%.1f" % x for x in ts[1])}]', flush=True)
print('ALL BIT-EQUAL' if ok_all else 'SOME DIFFERENT')
