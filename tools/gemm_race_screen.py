#!/usr/bin/env python
"""Race screen for the 256-tile GEMM's LDS-DMA ring (the guide's rule for any sync-structure edit: "screen it for races over many runs at
several sizes").  The kernel is deterministic (fixed K order per element), so ANY run whose output differs bitwise from the first run of the
same launch is a hazard (a fragment read that overtook its DMA, a slot re-filled under a reader).  Each shape runs REPS times, half of them
beside a bandwidth-heavy kernel on a second stream (uneven load moves the DMA landing times), with the epilogue forms the step uses; the
first output is also checked against torch (bf16 tolerance) so that "always the same wrong tile" cannot pass.

    python tools/gemm_race_screen.py [reps]        -> one line per shape, non-zero exit on any mismatch
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adapter4rec_amd import _lib as L

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(11)
r = lambda *s, sc=1.0, dt=torch.bfloat16: (torch.randn(*s, device=dev, generator=g) * sc).to(dt)
side = torch.cuda.Stream()
junk_a, junk_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
bad = 0
SHAPES = [(40448, 768, 768), (40448, 2304, 768), (40448, 3072, 768), (40448, 768, 3072), (40448, 768, 2304), (16896, 3072, 768), (16896, 768, 3072),
          (66304, 768, 768), (1280, 768, 3072), (2560, 3072, 768), (40448, 768, 128), (40448, 768, 192)]
for dt, tag in ((torch.bfloat16, 'bf16'), (torch.float32, 'fp32')):
    for (M, N, K) in SHAPES:
        if dt == torch.float32 and M * N > 40448 * 768:
            continue                                      # (the exact-fp32 MFMA is 16x slower: the small outputs only)
        A, B = r(M, K, dt=dt), r(N, K, sc=0.05, dt=dt)
        bias, R = r(N, sc=0.1, dt=torch.float32), r(M, N, dt=dt)
        forms = [dict(), dict(bias=bias, R1=R), dict(bias=bias, drop_p=0.1, drop_site=3, drop_seed=17, drop_first=True)]
        if dt == torch.bfloat16 and N % 256 == 0:
            forms.append('gelu8')
        for fi, f in enumerate(forms):
            first = first2 = None
            mism = 0
            for it in range(REPS):
                C = torch.empty(M, N, dtype=dt, device=dev)
                D = torch.empty(M, N, dtype=torch.uint8, device=dev) if f == 'gelu8' else None
                if it % 2:
                    with torch.cuda.stream(side):
                        junk_b.copy_(junk_a)                # 512 MB of HBM traffic beside the launch
                if f == 'gelu8':
                    L.gemm_nt(A, B, C, bias=bias, C2=D, act=L.ACT_GELU, c2_deriv='q8', q8_tiled=True)
                else:
                    L.gemm_nt(A, B, C, **f)
                torch.cuda.synchronize()
                if first is None:
                    first, first2 = C, D
                    ref = A.float() @ B.float().t()
                    if f == 'gelu8':
                        ref = torch.nn.functional.gelu(ref + bias)
                    elif 'bias' in f and 'R1' in f:
                        ref = ref + bias + R.float()
                    if not (isinstance(f, dict) and 'drop_p' in f):
                        err = float((C.float() - ref).abs().max() / ref.abs().max())
                        assert err < (2e-2 if dt == torch.bfloat16 else 1e-4), (tag, M, N, K, fi, err)
                elif not torch.equal(C, first) or (D is not None and not torch.equal(D, first2)):
                    mism += 1
            bad += mism
            print(f'{tag} M={M} N={N} K={K} form {fi}: {REPS} runs, {mism} differ from the first', flush=True)
print('RACE SCREEN', 'FAILED' if bad else 'clean', f'({bad} mismatching runs)')
sys.exit(1 if bad else 0)
