"""Coefficients and error table of csrc/a4r_common.h: gelu_poly_both_n (GELU + GELU' without exp / rcp for the bf16 + 8-bit-derivative epilogue).
Phi(x) - 0.5 = x P(t), phi(x) = G(t), t = x^2 / 8 - 1 on |x| <= 4; Chebyshev fits converted to monomials in t, evaluated as the kernel does
(fp32 Horner, x clamped to [-4, 4]) against float64 erf.       python tools/gelu_poly_fit.py

The device function it was fitted for is NOT in the library (it lost its A/B: profiles/r06_g_gelu_poly.txt).  Per pair of elements (packed fp32):
    xc = {v_med3_f32(x.x, -4, 4), v_med3_f32(x.y, -4, 4)};  xs = xc * 0.35355339;  t = xs * xs - 1
    pp = CP[8];  for k = 7 .. 0: pp = pp * t + CP[k]          gg = CG[7];  for k = 6 .. 0: gg = gg * t + CG[k]      (the two chains of four pairs interleaved)
    cdf = xc * pp + 0.5;   gelu = x * cdf;   gelu' = xc * gg + cdf"""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf

Phi = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))
phi = lambda x: np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def horner32(c, t):
    r = np.full_like(t, np.float32(c[-1]), dtype=np.float32)
    for a in c[-2::-1]:
        r = (r * t + np.float32(a)).astype(np.float32)
    return r


def main():
    tt = np.cos(np.pi * (np.arange(4001) + 0.5) / 4001)
    xx = np.sqrt(8 * (tt + 1))
    p1 = Ch.cheb2poly(Ch.chebfit(tt, (Phi(xx) - 0.5) / xx, 8))
    p2 = Ch.cheb2poly(Ch.chebfit(tt, phi(xx), 7))
    print('CP', ', '.join(repr(float(np.float32(a))) for a in p1))
    print('CG', ', '.join(repr(float(np.float32(a))) for a in p2))
    x = np.linspace(-8, 8, 400001).astype(np.float32)
    xc = np.clip(x, -4, 4).astype(np.float32)
    xs = (xc * np.float32(0.35355339059327373)).astype(np.float32)
    t = (xs * xs - np.float32(1)).astype(np.float32)
    cdf = (xc * horner32(p1, t) + np.float32(0.5)).astype(np.float32)
    g = (x * cdf).astype(np.float32)
    d = (xc * horner32(p2, t) + cdf).astype(np.float32)
    x64 = x.astype(np.float64)
    ge, de = x64 * Phi(x64), Phi(x64) + x64 * phi(x64)
    print('max |cdf error| %.2e   min cdf %.2e' % (np.abs(cdf - Phi(x64)).max(), cdf.min()))
    for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8)):
        m = (np.abs(x64) >= lo) & (np.abs(x64) <= hi)
        print('|x| in [%d, %d]: max |gelu error| %.2e   max |gelu\' error| %.2e (8-bit step 4.9e-3)' % (lo, hi, np.abs(g - ge)[m].max(), np.abs(d - de)[m].max()))


if __name__ == '__main__':
    main()
