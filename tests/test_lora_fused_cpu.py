"""a4r_lora_bwd_fused (include/a4r.h) restated on the CPU: the one-pass form must equal the five products it replaces (engine.py::_lora_backward_all,
the shared small-rank form: t = x A^T with a column of ones, dt = (dq B_q) s + (dv B_v) s, dB_. = d.^T t, dA = dt^T x) on the SAME scratch layout --
rank rows / columns 0 - 7 and 16 - 23 of the 64-wide padded rank dimension, the bias sums in column ONES_COL -- so that the corner flush that follows
is untouched.  (The HIP kernel against this restatement: tests/test_kernels_gpu.py::test_lora_bwd_fused.)"""
import torch

import sim_lib as S


import pytest


@pytest.mark.parametrize('r,R', [(8, 8), (12, 16), (15, 16)])
def test_fused_restatement_equals_the_five_products(r, R):
    g = torch.Generator().manual_seed(5)
    M, H, rp, oc = 256, 768, 64, 32
    T = torch.bfloat16
    x = torch.randn(M, H, generator=g).to(T)
    dqkv = (torch.randn(M, 3 * H, generator=g) * 0.1).to(T)
    dqa, dqb = dqkv[:, :H], dqkv[:, 2 * H:]
    A = torch.zeros(rp, H, dtype=T)
    A[0:r] = (torch.randn(r, H, generator=g) * 0.05).to(T)
    A[16:16 + r] = (torch.randn(r, H, generator=g) * 0.05).to(T)
    BTa, BTb = torch.zeros(rp, H, dtype=T), torch.zeros(rp, H, dtype=T)
    BTa[0:r] = (torch.randn(r, H, generator=g) * 0.05).to(T)
    BTb[16:16 + r] = (torch.randn(r, H, generator=g) * 0.05).to(T)
    sa, sb = 0.125, 0.25
    # the five products
    ones = torch.zeros(rp)
    ones[oc] = 1.0
    t, dt = torch.zeros(M, rp, dtype=T), torch.zeros(M, rp, dtype=T)
    S.gemm_nt(x, A, t, bias=ones, M=M)
    S.gemm_nt(dqa, BTa, dt, alpha=sa, M=M)
    S.gemm_nt(dqb, BTb, dt, alpha=sb, R1=dt, M=M)
    sBa, sBb, sA = torch.zeros(H, rp), torch.zeros(H, rp), torch.zeros(rp, H)
    S.gemm_tn2(dqa, t, sBa, dqb, t, sBb, M=M)
    S.gemm_tn(dt, x, sA, M=M)
    # the one pass
    fBa, fBb, fA = torch.zeros(H, rp), torch.zeros(H, rp), torch.zeros(rp, H)
    assert S.lora_bwd_fused_ok(x, M, H)
    S.lora_bwd_fused(x, dqa, dqb, A[0:R], A[16:16 + R], BTa[0:R], BTb[16:16 + R], sa, sb, fA[0:R], fA[16:16 + R], fBa[:, 0:R], fBb[:, 16:16 + R],
                     fBa[:, oc], fBb[:, oc], M, rank_rows=R)
    # what the corner flush reads: the rank corners and the ones column
    torch.testing.assert_close(fA[0:r], sA[0:r], rtol=2e-2, atol=2e-3)               # (dt holds the sum of two bf16-rounded products in the five-launch form)
    torch.testing.assert_close(fA[16:16 + r], sA[16:16 + r], rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(fBa[:, 0:r], sBa[:, 0:r], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fBb[:, 16:16 + r], sBb[:, 16:16 + r], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fBa[:, oc], sBa[:, oc], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fBb[:, oc], sBb[:, oc], rtol=1e-5, atol=1e-5)
    assert not S.lora_bwd_fused_ok(x.float(), M, H) and not S.lora_bwd_fused_ok(x, M, 384) and not S.lora_bwd_fused_ok(x, M + 8, H)
