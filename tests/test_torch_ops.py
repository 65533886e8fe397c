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


@pytest.mark.gpu
def test_layer_level_ops_equal_ctypes_binding():
    """The round-6 ops of SURVEY 8(b)'s list -- encoder_layer_fwd / _bwd, sasrec_block_fwd / _bwd, embed_ln_fwd, patch_embed_fwd, vit_assemble -- against the
    ctypes binding of the same C entry points (bit-equal outputs; fp32 atomic sums to their order)."""
    import math
    ops, _ = _ops()
    from adapter4rec_amd import _lib as L
    from test_kernels_gpu import _sasrec_case
    dev, t = 'cuda:0', torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(11)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    # ---- one encoder layer, forward and backward
    n_items, S, H, F, nh = 40, 30, 256, 512, 4
    M = 1280
    x = r(M, H).to(t)
    x[n_items * S:] = 0
    w = [r(3 * H, H, sc=0.05).to(t), r(3 * H, sc=0.1), r(H, H, sc=0.05).to(t), r(H, sc=0.1), r(F, H, sc=0.05).to(t), r(F, sc=0.1), r(H, F, sc=0.05).to(t), r(H, sc=0.1),
         1 + r(H, sc=0.1), r(H, sc=0.1), 1 + r(H, sc=0.1), r(H, sc=0.1)]
    ads = [[r(64, H, sc=0.05).to(t), r(64, sc=0.1), r(H, 64, sc=0.05).to(t), r(H, sc=0.1)] for _ in range(2)]
    mk = lambda c, dt=t: torch.zeros(M, c, dtype=dt, device=dev)
    km = torch.ones(n_items, S, device=dev)
    km[3, 20:] = 0
    scalars = dict(n_items=n_items, S=S, n_heads=nh, causal=False, scale=1 / math.sqrt(H // nh), mask_neg=float(torch.finfo(torch.float32).min), ln_eps=1e-12,
                   p_attn=0.1, p_hidden=0.1, drop_site=7, drop_seed=123, act1=L.ACT_GELU, act2=L.ACT_RELU)
    dx_fixed = r(M, H, sc=0.1).to(t)
    dx_fixed[n_items * S:] = 0
    runs = []
    for use_op in (True, False):
        saved = [mk(3 * H), mk(H), mk(H), mk(H), mk(64), mk(64), mk(F), mk(F, torch.uint8), mk(H), mk(H), mk(64), mk(64), mk(2, torch.float32), mk(2, torch.float32)]
        x1, xo = mk(H), mk(H)
        scratch = [mk(H), mk(H), mk(64), mk(H), mk(F), mk(H), mk(H), mk(3 * H)]
        grads = [[torch.zeros(H, 64, device=dev), torch.zeros(64, H, device=dev), torch.zeros(H, device=dev), torch.zeros(64, device=dev)] for _ in range(2)]
        dx_out, dx_in = dx_fixed, mk(H)
        wT = [w[0].t().contiguous(), w[2].t().contiguous(), w[4].t().contiguous(), w[6].t().contiguous()]
        adT = [[a[0].t().contiguous(), a[2].t().contiguous()] for a in ads]
        if use_op:
            ops.encoder_layer_fwd(x, w, ads[0], ads[1], saved, x1, xo, km, None, q8_tiled=False, **scalars)
            ops.encoder_layer_bwd(dx_out, x1, xo, w, wT, ads[0], ads[1], adT[0], adT[1], saved, scratch, grads[0], grads[1], dx_in, km, None, q8_tiled=False, **scalars)
        else:
            d = L.EncoderLayer()
            d.M, d.H, d.F, d.n_items, d.S, d.n_heads, d.dh, d.causal = M, H, F, n_items, S, nh, H // nh, 0
            d.scale, d.mask_neg, d.ln_eps, d.p_attn, d.p_hidden, d.drop_site, d.drop_seed = scalars['scale'], scalars['mask_neg'], 1e-12, 0.1, 0.1, 7, 123
            d.key_mask = km.data_ptr()
            for name, ten in zip(('wqkv', 'bqkv', 'wo', 'bo', 'wi', 'bi', 'wo2', 'bo2', 'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b'), w):
                setattr(d, name, ten.data_ptr())
            for name, ten in zip(('wqkvT', 'woT', 'wiT', 'wo2T'), wT):
                setattr(d, name, ten.data_ptr())
            for k in range(2):
                a = d.ad[k]
                a.wd, a.bd, a.wu, a.bu, a.wdT, a.wuT = (q.data_ptr() for q in (ads[k][0], ads[k][1], ads[k][2], ads[k][3], adT[k][0], adT[k][1]))
                a.act = scalars['act1'] if k == 0 else scalars['act2']
                a.g_wu, a.g_wd, a.g_bu, a.g_bd, a.ldg_wu, a.ldg_wd = grads[k][0].data_ptr(), grads[k][1].data_ptr(), grads[k][2].data_ptr(), grads[k][3].data_ptr(), 64, H
            for name, ten in zip(('qkv', 'ctx', 'h1', 'v1', 'zp1', 'z1', 'u', 'upre', 'h2', 'v2', 'zp2', 'z2', 'st1', 'st2'), saved):
                setattr(d, name, ten.data_ptr())
            d.upre_q8 = 1
            for name, ten in zip(('dv1', 'dv2', 'dzp', 'd_h', 'du', 'dx1', 'dctx', 'dqkv'), scratch):
                setattr(d, name, ten.data_ptr())
            L.encoder_layer_fwd(d, x, x1, xo)
            L.encoder_layer_bwd(d, x1, xo, dx_out, dx_in)
        runs.append(dict(x1=x1, xo=xo, dx_in=dx_in, grads=grads, dx_out=dx_out))
    assert torch.equal(runs[0]['x1'], runs[1]['x1']) and torch.equal(runs[0]['xo'], runs[1]['xo']) and torch.equal(runs[0]['dx_in'], runs[1]['dx_in'])
    for k in range(2):
        for ga, gb in zip(runs[0]['grads'][k], runs[1]['grads'][k]):
            torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-5 * float(gb.abs().max()))
    assert float(runs[0]['xo'].float().abs().max()) > 0.1 and float(runs[0]['dx_in'].float().abs().max()) > 0
    assert all(float(gr.abs().max()) > 0 for k in range(2) for gr in runs[0]['grads'][k])
    # ---- SASRec block
    desc, xs, mask, dy = _sasrec_case(16, 1, True, seed=5, mode=0)
    B, T = 6, 20
    y0, y1 = torch.zeros_like(xs), torch.zeros_like(xs)
    L.sasrec_block(desc, xs, mask, y0, B, T, False)
    names = ('wqkv', 'wfc', 'w1', 'b1', 'w2', 'b2', 'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b', 'wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2')
    wl = [desc[n] for n in names]
    common = dict(n_heads=desc['n_heads'], F=desc['F'], d=desc['d'], act=desc['act'], inner_res=bool(desc['inner_res']), eps=desc['eps'], mask_neg=desc['mask_neg'])
    ops.sasrec_block_fwd(xs.view(B, T, 64), mask.view(B, T), y1.view(B, T, 64), wl, **common)
    assert torch.equal(y0, y1)
    gl = [torch.zeros_like(desc['g_' + n]) for n in ('wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2')]
    dx0, dx1 = torch.zeros_like(xs), torch.zeros_like(xs)
    L.sasrec_block(desc, xs, mask, dx0, B, T, False, dy=dy)
    ops.sasrec_block_bwd(xs.view(B, T, 64), mask.view(B, T), dy.view(B, T, 64), dx1.view(B, T, 64), wl, gl, **common)
    assert torch.equal(dx0, dx1)
    for n, gnew in zip(('wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2'), gl):
        torch.testing.assert_close(gnew, desc['g_' + n], rtol=1e-4, atol=1e-5 * float(desc['g_' + n].abs().max()) + 1e-9)
    # ---- embeddings
    V, Hh, n, S2 = 500, 256, 24, 30
    ids = torch.randint(1, V, (n, 2 * S2), device=dev, generator=g)
    word, pos, typ, gam, bet = r(V, Hh, sc=0.1), r(64, Hh, sc=0.1), r(Hh, sc=0.1), 1 + r(Hh, sc=0.1), r(Hh, sc=0.1)
    Mp = 768
    e0, e1 = torch.zeros(Mp, Hh, dtype=t, device=dev), torch.zeros(Mp, Hh, dtype=t, device=dev)
    L.embed_ln(ids, word, pos, typ, gam, bet, 1e-12, e0, n, S2, drop_p=0.1, drop_site=3, drop_seed=9)
    ops.embed_ln_fwd(ids, word, pos, typ, gam, bet, 1e-12, e1, S2, False, 0, 0.1, 3, 9)
    assert torch.equal(e0, e1) and float(e0.float().abs().max()) > 0
    # ---- patch embedding input side
    img = torch.randint(0, 256, (3, 32, 32, 3), dtype=torch.uint8, device=dev, generator=g)
    p0, p1 = torch.zeros(128, 768, dtype=t, device=dev), torch.zeros(128, 768, dtype=t, device=dev)
    L.patchify(img, p0, 16)
    ops.patch_embed_fwd(img, p1, 16)
    assert torch.equal(p0, p1) and float(p0.float().abs().max()) > 0
    with pytest.raises(RuntimeError, match='invalid argument'):
        ops.encoder_layer_fwd(x, w, ads[0], ads[1], saved, x1, xo, None, None, n_items, 64, nh, False, 0.1, -1e9, 1e-12)        # 64 tokens: outside the layer call's scope


@pytest.mark.gpu
def test_autograd_functions_over_the_layer_ops():
    """torch_ops.EncoderLayerFunction / SasrecBlockFunction (round 6: the autograd forms of encoder_layer_fwd / _bwd and sasrec_block_fwd / _bwd).
    (a) the encoder layer against torch autograd of a plain fp32 restatement of the same layer (HF BertLayer + serial Houlsby adapters, dropout off):
        output within 3e-2, d x and every adapter gradient within 6 % of its tensor's max (bf16 storage against fp32);
    (b) the SASRec block against the out-variant ops called directly (bit-equal d x, gradients to atomic order)."""
    import math
    ops, _ = _ops()
    from adapter4rec_amd import _lib as L
    from adapter4rec_amd import torch_ops
    from test_kernels_gpu import _sasrec_case
    dev, t = 'cuda:0', torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(21)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    n_items, S, H, F, nh = 40, 30, 256, 512, 4
    M = 1280
    x = r(M, H).to(t)
    x[n_items * S:] = 0
    frozen = [r(3 * H, H, sc=0.05).to(t), r(3 * H, sc=0.1), r(H, H, sc=0.05).to(t), r(H, sc=0.1), r(F, H, sc=0.05).to(t), r(F, sc=0.1), r(H, F, sc=0.05).to(t), r(H, sc=0.1),
              1 + r(H, sc=0.1), r(H, sc=0.1), 1 + r(H, sc=0.1), r(H, sc=0.1)]
    ads = [r(64, H, sc=0.05), r(64, sc=0.1), r(H, 64, sc=0.05), r(H, sc=0.1), r(64, H, sc=0.05), r(64, sc=0.1), r(H, 64, sc=0.05), r(H, sc=0.1)]
    km = torch.ones(n_items, S, device=dev)
    km[3, 20:] = 0
    cfg = dict(n_items=n_items, S=S, n_heads=nh, act1=L.ACT_GELU, act2=L.ACT_GELU)
    dy = r(M, H, sc=0.1)
    dy[n_items * S:] = 0
    xa = x.clone().requires_grad_(True)
    pa = [p.clone().requires_grad_(True) for p in ads]
    y = torch_ops.EncoderLayerFunction.apply(xa, *pa, frozen, km, cfg)
    y.backward(dy.to(t))
    # fp32 restatement
    n = n_items * S
    xr = x[:n].float().clone().requires_grad_(True)
    pr = [p.clone().requires_grad_(True) for p in ads]
    W = [q.float() for q in frozen]
    gelu = torch.nn.functional.gelu
    qkv = xr @ W[0].t() + W[1]
    q, k, v = (z.view(n_items, S, nh, H // nh).transpose(1, 2) for z in qkv.split(H, 1))
    sc = q @ k.transpose(-1, -2) / math.sqrt(H // nh) + (1 - km)[:, None, None, :] * -1e9
    ctxr = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(n, H)
    h1 = ctxr @ W[2].t() + W[3]
    a1 = gelu(h1 @ pr[0].t() + pr[1]) @ pr[2].t() + pr[3] + h1
    x1 = torch.nn.functional.layer_norm(a1 + xr, (H,), W[8], W[9], 1e-12)
    u = gelu(x1 @ W[4].t() + W[5])
    h2 = u @ W[6].t() + W[7]
    a2 = gelu(h2 @ pr[4].t() + pr[5]) @ pr[6].t() + pr[7] + h2
    yr = torch.nn.functional.layer_norm(a2 + x1, (H,), W[10], W[11], 1e-12)
    yr.backward(dy[:n])
    assert float((y[:n].float() - yr).abs().max()) < 3e-2 * max(1.0, float(yr.abs().max()))
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
    errs = {'dx': rel(xa.grad[:n], xr.grad)}
    for i, (a, b) in enumerate(zip(pa, pr)):
        assert a.grad is not None and a.grad.shape == b.grad.shape and a.grad.dtype == torch.float32
        errs[f'p{i}'] = rel(a.grad, b.grad)
    print('EncoderLayerFunction vs fp32 torch autograd:', {k: round(v, 4) for k, v in errs.items()})
    assert max(errs.values()) < 0.06, errs
    # ---- SASRec block
    desc, xs, mask, dys = _sasrec_case(16, 1, True, seed=5, mode=0)
    B, T = 6, 20
    fz = tuple(desc[n_] for n_ in ('wqkv', 'wfc', 'w1', 'b1', 'w2', 'b2', 'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b'))
    adn = ('wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2')
    adp = [desc[n_].clone().requires_grad_(True) for n_ in adn]
    cfg2 = dict(n_heads=desc['n_heads'], F=desc['F'], d=desc['d'], act=desc['act'], inner_res=bool(desc['inner_res']), eps=desc['eps'], mask_neg=desc['mask_neg'])
    xq = xs.view(B, T, 64).clone().requires_grad_(True)
    yq = torch_ops.SasrecBlockFunction.apply(xq, mask.view(B, T), fz, *adp, cfg2)
    yq.backward(dys.view(B, T, 64))
    wl = [*fz, *[desc[n_] for n_ in adn]]
    y0, dx0 = torch.zeros_like(xs), torch.zeros_like(xs)
    gl = [torch.zeros_like(desc[n_]) for n_ in adn]
    ops.sasrec_block_fwd(xs.view(B, T, 64), mask.view(B, T), y0.view(B, T, 64), wl, **cfg2)
    ops.sasrec_block_bwd(xs.view(B, T, 64), mask.view(B, T), dys.view(B, T, 64), dx0.view(B, T, 64), wl, gl, **cfg2)
    assert torch.equal(yq.detach().reshape(-1), y0.reshape(-1)) and torch.equal(xq.grad.reshape(-1), dx0.reshape(-1))
    for a, b in zip(adp, gl):
        torch.testing.assert_close(a.grad, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-9)
