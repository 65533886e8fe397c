"""Time a4r_adapter_fwd against the unfused sequence (down GEMM, up GEMM + residuals, LayerNorm) on one GPU."""
import sys, torch
sys.path.insert(0, '.')
from adapter4rec_amd import _lib as L

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

M, H = int(sys.argv[1]) if len(sys.argv) > 1 else 40448, 768
d = 'cuda'
g = torch.Generator(device=d).manual_seed(0)
bf = torch.bfloat16
h = torch.randn(M, H, device=d, generator=g).to(bf); x = torch.randn(M, H, device=d, generator=g).to(bf)
Wd = (torch.randn(64, H, device=d, generator=g) * .05).to(bf); Wu = (torch.randn(H, 64, device=d, generator=g) * .05).to(bf)
bd = torch.randn(64, device=d) * .1; bu = torch.randn(H, device=d) * .1
gam = torch.rand(H, device=d) + .5; bet = torch.randn(H, device=d) * .1
zp = torch.empty(M, 64, device=d, dtype=bf); z = torch.empty_like(zp)
v = torch.empty(M, H, device=d, dtype=bf); y = torch.empty_like(v); st = torch.empty(M, 2, device=d)
def fused(): L.adapter_fwd(h, x, Wd, bd, Wu, bu, gam, bet, 1e-12, 1, True, zp, z, v, y, st)
def unfused():
    L.gemm_nt(h, Wd, z, bias=bd, C2=zp, act=1)
    L.gemm_nt(z, Wu, v, bias=bu, R1=h, R2=x)
    L.ln_fwd(v, gam, bet, 1e-12, y, st)
tf, tu = timeit(fused), timeit(unfused)
byt = 4 * M * H * 2
print(f"M={M} fused {tf:.1f} us ({byt / tf / 1e6:.2f} TB/s algorithmic)  unfused {tu:.1f} us")
