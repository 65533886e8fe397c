"""GPU check of the four-wave GEMM experiment (moved out of tests/ with the kernel in round 6: the shipped library does not contain it).

  make -C adapter4rec_amd/csrc W4=1 -j8 && A4R_LIB_PATH=tools/_ab/liba4r_w4.so python -m pytest tools/w4/check_bit_equal.py -q
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_kernels_gpu import dev, rnd  # noqa: E402


@pytest.mark.parametrize('N,K', [(768, 768), (768, 3072), (2304, 256), (3072, 768)])
def test_gemm256_four_wave_bit_equal_to_eight_wave(N, K):
    """a4r_gemm_variant 9: the four-wave, hand-scheduled form of the 256-tile kernel (a4r_gemm256w4.hip; K loop = generated asm text, the epilogue
    = the eight-wave kernel's text on a different accumulator layout).  Same K order per element and the same epilogue arithmetic: every form of
    the training step must come out BIT-EQUAL to the eight-wave kernel (variant 8) -- outputs, second outputs, dropout patterns, and the tile-native
    8-bit derivative written by one kernel and read by the other.  180 row panels: every workgroup walks 2 - 9 tiles (the unit stream across tiles,
    staggered starts).  Also against fp32 torch on the plain form."""
    from adapter4rec_amd import _lib as L
    t = torch.bfloat16
    M = 180 * 256
    A, B = rnd(M, K, dtype=t, seed=91), rnd(N, K, dtype=t, scale=0.05, seed=92)
    bias, R1, Pre = rnd(N, seed=93), rnd(M, N, dtype=t, seed=94), rnd(M, N, dtype=t, seed=95)
    res = {}
    try:
        for v in (8, 9):
            L.gemm_variant(v)
            o = {k: torch.full((M, N), float('nan'), dtype=t, device=dev()) for k in ('plain', 'drop', 'res', 'dropres', 'gelu8t', 'gelu', 'gelu_c2', 'dmul', 'relu')}
            C8 = torch.zeros(M, N, dtype=torch.uint8, device=dev())
            f32 = torch.zeros(M, N, dtype=torch.float32, device=dev())
            L.gemm_nt(A, B, o['plain'], bias=bias)
            L.gemm_nt(A, B, o['drop'], bias=bias, drop_p=0.1, drop_site=3, drop_seed=11)
            L.gemm_nt(A, B, o['res'], R1=R1)
            L.gemm_nt(A, B, o['dropres'], bias=bias, R1=R1, drop_p=0.1, drop_site=5, drop_seed=13, drop_first=True)
            L.gemm_nt(A, B, o['gelu8t'], bias=bias, C2=C8, act=L.ACT_GELU, c2_deriv='q8', q8_tiled=True)
            res[(v, 'C8')] = C8
            L.gemm_nt(A, B, o['gelu'], bias=bias, C2=o['gelu_c2'], act=L.ACT_GELU, c2_deriv=True)
            L.gemm_nt(A, B, o['dmul'], Pre=Pre, dact=L.DACT_MUL)
            L.gemm_nt(A, B, o['relu'], bias=bias, act=L.ACT_RELU, alpha=0.5)
            L.gemm_nt(A, B, f32, bias=bias)
            res[(v, 'f32')] = f32
            res[v] = o
        # the derivative tensor written by the EIGHT-wave kernel, read by the FOUR-wave kernel and the other way round: one byte layout
        for w, r in ((8, 9), (9, 8)):
            L.gemm_variant(r)
            out = torch.full((M, N), float('nan'), dtype=t, device=dev())
            L.gemm_nt(A, B, out, Pre=res[(w, 'C8')], dact=L.DACT_MUL_Q8, q8_tiled=True)
            res[(w, r, 'dmul8t')] = out
    finally:
        L.gemm_variant(8)
    for k in res[8]:
        assert torch.equal(res[8][k].view(torch.int16), res[9][k].view(torch.int16)), k
        assert not bool(res[9][k].isnan().any()), k
    assert torch.equal(res[(8, 'C8')], res[(9, 'C8')])
    assert torch.equal(res[(8, 'f32')], res[(9, 'f32')])
    assert torch.equal(res[(8, 9, 'dmul8t')].view(torch.int16), res[(9, 8, 'dmul8t')].view(torch.int16))
    close(res[9]['plain'], A.float() @ B.float().t() + bias, t, 'four-wave plain vs torch')
