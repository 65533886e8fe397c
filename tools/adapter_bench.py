"""Time a4r_adapter_ln_fwd / _bwd against the three-launch forms they replace, at the training step's shape
(M = 40 448 rows x H = 768, bottleneck 64, bf16), back to back on one GPU.  Prints us per launch and achieved TB/s on the
algorithmic bytes ((4 H + 128) * 2 * M).  usage: python tools/adapter_bench.py [M] [H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adapter4rec_amd import _lib as L

M = int(sys.argv[1]) if len(sys.argv) > 1 else 40448
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev, t = 'cuda:0', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
h, x = r(M, H).to(t), r(M, H).to(t)
Wd, Wu = r(64, H, sc=0.05).to(t), r(H, 64, sc=0.05).to(t)
WuT, WdT = Wu.t().contiguous(), Wd.t().contiguous()
bd, bu, gam, bet = r(64, sc=0.1), r(H, sc=0.1), 1 + r(H, sc=0.1), r(H, sc=0.1)
mk = lambda c: torch.zeros(M, c, dtype=t, device=dev)
zp, z, v, y, st = mk(64), mk(64), mk(H), mk(H), torch.zeros(M, 2, device=dev)
dy, dv, dzp, dh, dbias = r(M, H).to(t), mk(H), mk(64), mk(H), torch.zeros(H, device=dev)


def fwd_fused():
    L.adapter_ln_fwd(h, h, x, Wd, bd, Wu, bu, gam, bet, 1e-12, 1, zp, z, v, y, st)


def fwd_fused_y():          # the training step's form: only y = LN(v) is kept (backward rebuilds xhat from it)
    L.adapter_ln_fwd(h, h, x, Wd, bd, Wu, bu, gam, bet, 1e-12, 1, zp, z, None, y, st)


def bwd_fused_y():          # the step's form: xhat from y; the bias gradients ride in the weight-gradient launch (no column sums here)
    L.adapter_ln_bwd(dy, y, st, gam, None, zp, 1, WuT, WdT, True, dv, dzp, dh, drop_p=0.1, drop_site=3, drop_seed=7, beta_y=bet)


def fwd_three():
    L.gemm_nt(h, Wd, z, bias=bd, C2=zp, act=1)
    L.gemm_nt(z, Wu, v, bias=bu, R1=h, R2=x)
    L.ln_fwd(v, gam, bet, 1e-12, y, st)


def bwd_fused():
    L.adapter_ln_bwd(dy, v, st, gam, None, zp, 1, WuT, WdT, True, dv, dzp, dh, dbias=dbias, drop_p=0.1, drop_site=3, drop_seed=7)


def bwd_three():
    L.ln_bwd(dy, v, st, gam, dv, dbias=dbias)
    L.gemm_nt(dv, WuT, dzp, Pre=zp, dact=1)
    L.gemm_nt(dzp, WdT, dh, R1=dv, drop_p=0.1, drop_site=3, drop_seed=7)


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd_three()
alg = (4 * H + 128) * 2 * M
alg_y = (3 * H + 128) * 2 * M
for name, f, b in (('fwd fused (step)', fwd_fused_y, alg_y), ('bwd fused (step)', bwd_fused_y, alg)):
    us = timeit(f)
    print(f'{name:16s} {us:8.1f} us   {b / us / 1e6:6.2f} TB/s on {b / 1e6:.0f} MB')
for name, f in (('fwd fused', fwd_fused), ('fwd 3 launches', fwd_three), ('bwd fused', bwd_fused), ('bwd 3 launches', bwd_three)):
    us = timeit(f)
    print(f'{name:16s} {us:8.1f} us   {alg / us / 1e6:6.2f} TB/s on the fused form\'s algorithmic bytes ({alg / 1e6:.0f} MB)')
