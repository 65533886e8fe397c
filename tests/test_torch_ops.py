"""The TORCH_LIBRARY op layer (adapter4rec_amd/csrc/a4r_torch_ops.cpp): registered schemas and TORCH_CHECK errors without a
GPU; on the GPU every op against the ctypes binding of the same C entry point (bit-equal: it IS the same kernel launch)."""
import pytest
import torch


def _ops():
    from adapter4rec_amd import torch_ops
    return torch_ops.load(), torch_ops


def test_ops_registered_and_abi_version():
    ops, mod = _ops()
    from adapter4rec_amd import _lib
    assert int(ops.abi_version()) == _lib.ABI_VERSION
    for name in mod.OPS:
        schema = str(getattr(ops, name).default._schema)
        assert schema.startswith('a4r::' + name + '('), schema
    # outputs are caller-allocated and declared as mutated in the schema (functionalisation / torch.compile see the aliasing)
    assert 'Tensor(a!) C' in str(ops.gemm_nt.default._schema)


def test_torch_check_errors_without_gpu():
    """Device / shape / dtype mistakes raise RuntimeError from TORCH_CHECK (never abort, never a silent CPU path)."""
    ops, _ = _ops()
    a, b, c = torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match='must be a device tensor'):
        ops.gemm_nt(a, b, c)
    with pytest.raises(RuntimeError, match='must be a device tensor'):
        ops.ln_fwd(a, None, torch.ones(64), torch.zeros(64), 1e-12, c, torch.zeros(128, 2))
    with pytest.raises(RuntimeError):
        ops.topk_rank_eval(torch.zeros(4, 8), torch.zeros(9, 7), torch.zeros(4, dtype=torch.int32), torch.zeros(5, dtype=torch.int32),
                           torch.zeros(1, dtype=torch.int32), torch.zeros(4, dtype=torch.int32))


@pytest.mark.gpu
def test_ops_equal_ctypes_binding():
    ops, _ = _ops()
    from adapter4rec_amd import _lib as L
    dev, t = 'cuda:0', torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    M, H, d = 1024, 768, 64
    # ---- gemm_nt with bias, residual, dropout and GELU
    A, W, R, bias = r(M, H).to(t), r(H, H, sc=0.05).to(t), r(M, H).to(t), r(H, sc=0.1)
    c0, c1 = torch.zeros(M, H, dtype=t, device=dev), torch.zeros(M, H, dtype=t, device=dev)
    L.gemm_nt(A, W, c0, bias=bias, R1=R, act=L.ACT_GELU, drop_p=0.1, drop_site=7, drop_seed=99, drop_first=True)
    ops.gemm_nt(A, W, c1, bias, R, None, L.ACT_GELU, 1.0, 0.1, 7, 99, True)
    assert torch.equal(c0, c1) and float(c0.float().abs().max()) > 0
    with pytest.raises(RuntimeError, match='B must be'):
        ops.gemm_nt(A, W[:, :64], c1)
    with pytest.raises(RuntimeError, match='invalid argument'):
        ops.gemm_nt(A[:100], W, c1[:100])                       # M % 128 != 0: the C ABI's status code surfaces as an exception
    # ---- fused adapter forward / backward
    Wd, Wu, bd, bu = r(d, H, sc=0.05).to(t), r(H, d, sc=0.05).to(t), r(d, sc=0.1), r(H, sc=0.1)
    gam, bet = 1 + r(H, sc=0.1), r(H, sc=0.1)
    mk = lambda c, dt=t: torch.zeros(M, c, dtype=dt, device=dev)
    out0 = [mk(d), mk(d), mk(H), mk(H), torch.zeros(M, 2, device=dev), mk(H, torch.float32)]
    out1 = [mk(d), mk(d), mk(H), mk(H), torch.zeros(M, 2, device=dev), mk(H, torch.float32)]
    res32 = R.float()
    L.adapter_ln_fwd(A, A, R, Wd, bd, Wu, bu, gam, bet, 1e-12, L.ACT_RELU, *out0[:5], res32=res32, y32=out0[5])
    ops.adapter_residual_ln_fwd(A, A, R, Wd, bd, Wu, bu, gam, bet, 1e-12, L.ACT_RELU, *out1[:5], res32, out1[5])
    for x, y in zip(out0, out1):
        assert torch.equal(x, y)
    assert float((out0[5] - out0[3].float()).abs().max()) < 0.05          # y32 is y before its bf16 rounding
    dy = r(M, H).to(t)
    WuT, WdT = Wu.t().contiguous(), Wd.t().contiguous()
    b0 = [mk(H), mk(d), mk(H), torch.zeros(H, device=dev), torch.zeros(d, device=dev)]
    b1 = [mk(H), mk(d), mk(H), torch.zeros(H, device=dev), torch.zeros(d, device=dev)]
    L.adapter_ln_bwd(dy, out0[2], out0[4], gam, None, out0[0], L.ACT_RELU, WuT, WdT, True, b0[0], b0[1], b0[2], dbias=b0[3], dbd=b0[4])
    ops.adapter_residual_ln_bwd(dy, out0[2], out0[4], gam, None, out0[0], L.ACT_RELU, WuT, WdT, True, b1[0], b1[1], b1[2], None, None, b1[3], b1[4])
    for x, y in zip(b0[:3], b1[:3]):
        assert torch.equal(x, y)
    for x, y in zip(b0[3:], b1[3:]):                                      # column sums: fp32 atomics, order-dependent in the last bits
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-4 * float(x.abs().max()))
    # ---- LayerNorm
    y0, y1, s0, s1 = mk(H), mk(H), torch.zeros(M, 2, device=dev), torch.zeros(M, 2, device=dev)
    L.ln_fwd(A, gam, bet, 1e-12, y0, s0)
    ops.ln_fwd(A, None, gam, bet, 1e-12, y1, s1)
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    # ---- Adam over a flat buffer with two lr groups
    n = 4096
    p0 = r(n); p1 = p0.clone(); gr = r(n)
    m0, v0, m1, v1 = (torch.zeros(n, device=dev) for _ in range(4))
    seg_end = torch.tensor([1024, 4096], dtype=torch.int32, device=dev)
    seg_grp = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    lrs = torch.tensor([1e-3, 1e-2], device=dev)
    ops.fused_adam_step(p1, gr, m1, v1, seg_end, seg_grp, lrs, 1)
    ref = torch.optim.Adam([{'params': [torch.nn.Parameter(p0[:1024].clone())], 'lr': 1e-3}, {'params': [torch.nn.Parameter(p0[1024:].clone())], 'lr': 1e-2}])
    for grp, sl in zip(ref.param_groups, (slice(0, 1024), slice(1024, n))):
        grp['params'][0].grad = gr[sl].clone()
    ref.step()
    want = torch.cat([grp['params'][0].detach() for grp in ref.param_groups])
    torch.testing.assert_close(p1, want, rtol=2e-5, atol=1e-7)
    # ---- eval rank: U users, N items, CSR history
    U, N1, E = 16, 501, 64
    prec, emb = r(U, E), r(N1, E)
    target = torch.randint(1, N1, (U,), device=dev, generator=g, dtype=torch.int32)
    hist_ptr = torch.arange(0, 3 * U + 1, 3, dtype=torch.int32, device=dev)
    hist_idx = torch.randint(1, N1, (3 * U,), device=dev, generator=g, dtype=torch.int32)
    rank = torch.zeros(U, dtype=torch.int32, device=dev)
    ops.topk_rank_eval(prec, emb, target, hist_ptr, hist_idx, rank)
    sc = prec @ emb.t()
    for u in range(U):
        s = sc[u].clone()
        tu = int(target[u])
        hist = [int(x) for x in hist_idx[3 * u:3 * u + 3] if int(x) != tu]
        s[hist] = float('-inf')
        s[0] = float('-inf')
        assert int(rank[u]) == 1 + int((s > s[tu]).sum()), u
    # ---- LoRA backward in one pass (ranks 8 + 5 in the shared-tile form) against the ctypes binding
    Ml = 16 * 300
    xl, dqkv = r(Ml, H).to(t), r(Ml, 3 * H, sc=0.1).to(t)
    dqa, dqb = dqkv[:, :H], dqkv[:, 2 * H:]
    Aop, BTa, BTb = (torch.zeros(64, H, dtype=t, device=dev) for _ in range(3))
    Aop[0:8], Aop[16:21], BTa[0:8], BTb[16:21] = r(8, H, sc=0.05).to(t), r(5, H, sc=0.05).to(t), r(8, H, sc=0.05).to(t), r(5, H, sc=0.05).to(t)
    res = []
    for use_op in (False, True):
        sA, sBa, sBb = torch.zeros(64, H, device=dev), torch.zeros(H, 64, device=dev), torch.zeros(H, 64, device=dev)
        if use_op:
            ws = torch.empty(int(L.lib().a4r_lora_bwd_fused_ws_floats(H)), device=dev)
            ops.lora_bwd(xl, dqa, dqb, Aop[0:8], Aop[16:24], BTa[0:8], BTb[16:24], 0.125, 0.25, sA[0:8], sA[16:24], sBa[:, 0:8], sBb[:, 16:24], sBa[:, 32], sBb[:, 32], ws)
        else:
            L.lora_bwd_fused(xl, dqa, dqb, Aop[0:8], Aop[16:24], BTa[0:8], BTb[16:24], 0.125, 0.25, sA[0:8], sA[16:24], sBa[:, 0:8], sBb[:, 16:24], sBa[:, 32], sBb[:, 32], Ml)
        res.append((sA, sBa, sBb))
    for a_, b_ in zip(*res):                                  # the same two launches; the second one's atomics may land in another order
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-5 * float(b_.abs().max()))
    assert float(res[1][0].abs().max()) > 0 and float(res[1][1][:, 32].abs().max()) > 0
    with pytest.raises(RuntimeError, match='rank rows'):
        ops.lora_bwd(xl, dqa, dqb, Aop[0:4], Aop[16:20], BTa[0:4], BTb[16:20], 1.0, 1.0, sA[0:4], sA[16:20], sBa[:, 0:8], sBb[:, 16:24], None, None, ws)
