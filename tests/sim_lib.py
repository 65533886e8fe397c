"""TEST-ONLY stand-in for adapter4rec_amd._lib: the same wrapper signatures, executed with plain torch on the CPU.

Purpose: exercise the HOST logic of adapter4rec_amd/engine.py (buffer plumbing, launch order, gradient routing,
optimizer binding) in the build container, which has no GPU.  It follows the semantics written in include/a4r.h.
It is never imported by the package; tests monkeypatch it in.  Dropout must be off (p = 0).
"""
import ctypes
import math

import torch

from adapter4rec_amd import _lib as REAL

BF16, F32, FP8 = REAL.BF16, REAL.F32, REAL.FP8
quantize_weight_fp8 = REAL.quantize_weight_fp8
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_TANH, ACT_LEAKY = 0, 1, 2, 3, 4
DACT_MUL = 15
DACT_MUL_Q8 = 14
Q8_OFF, Q8_STEP = REAL.Q8_OFF, REAL.Q8_STEP
EVAL_MAX_HISTORY = REAL.EVAL_MAX_HISTORY
ACT_BY_NAME = REAL.ACT_BY_NAME
PackDesc, PhmDesc, AddDesc, desc_table = REAL.PackDesc, REAL.PhmDesc, REAL.AddDesc, REAL.desc_table


def lib():
    return None


def require_gpu(*t):
    return None


def _act(x, a):
    if a == 1:
        return torch.relu(x)
    if a == 2:
        return torch.nn.functional.gelu(x)
    if a == 3:
        return torch.nn.functional.gelu(x, approximate='tanh')
    if a == 4:
        return torch.nn.functional.leaky_relu(x, 0.01)
    return x


def _dact(pre, a):
    with torch.enable_grad():        # callers may sit inside an autograd.Function.backward (grad mode off)
        p = pre.detach().clone().requires_grad_(True)
        _act(p, a).sum().backward()
    return p.grad


def _deq(t, scale):
    """e4m3 bit patterns (uint8) + per-row scale -> fp32."""
    return t.view(torch.float8_e4m3fn).float() * scale.reshape(-1)[:t.shape[0], None]


def gemm_nt(A, B, Cout, bias=None, C2=None, R1=None, R2=None, Pre=None, act=0, dact=0, alpha=1.0,
            drop_p=0.0, drop_site=0, drop_seed=0, M=None, drop_first=False, c2_deriv=False, scale_a=None, scale_b=None,
            c_fp8=0, c_scale=1.0, c_scale_out=None, q8_tiled=False):
    assert drop_p == 0.0          # (q8_tiled: a storage order private to the two launches that share the tensor -- nothing to simulate)
    M = A.shape[0] if M is None else M
    assert M % 128 == 0 and B.shape[0] % 64 == 0 and B.shape[1] % 64 == 0, (M, B.shape)
    if A.dtype == torch.uint8:           # fp8 operands (include/a4r.h: A4R_FP8)
        assert B.dtype == torch.uint8 and M % 256 == 0 and B.shape[0] % 256 == 0 and B.shape[1] % 128 == 0 and (dact == 0 or (dact == 14 and c_fp8 == 2))
        v = alpha * (_deq(A[:M], scale_a) @ _deq(B, scale_b).t())
    else:
        v = alpha * (A[:M].float() @ B.float().t())
    if bias is not None:
        v = v + bias
    if C2 is not None and c2_deriv == 'q8':      # include/a4r.h c2_mode 2: gelu' as 8-bit fixed point
        assert act == ACT_GELU and C2.dtype == torch.uint8 and (Cout.dtype == torch.bfloat16 or c_fp8)
        C2[:M] = torch.clamp(torch.round((_dact(v, act) + Q8_OFF) * (1.0 / Q8_STEP)), 0, 255).to(torch.uint8)
    elif C2 is not None:
        C2[:M] = (_dact(v, act) if c2_deriv else v).to(C2.dtype)
    v = _act(v, act)
    if dact == 14:
        assert Pre.dtype == torch.uint8
        v = v * (Pre[:M].float() * Q8_STEP - Q8_OFF)
    elif dact == 15:
        v = v * Pre[:M].float()
    elif dact:
        v = v * _dact(Pre[:M].float(), dact)
    if R1 is not None:
        v = v + R1[:M].float()
    if R2 is not None:
        v = v + R2[:M].float()
    if c_fp8:                            # e4m3 output (include/a4r.h a4r_gemm_t.c_fp8): static scale, or the A row's scale x c_scale
        assert A.dtype == torch.uint8 and Cout.dtype == torch.uint8 and R1 is None and R2 is None
        so = torch.full((M,), float(c_scale)) if c_fp8 == 1 else scale_a.reshape(-1)[:M].float() * float(c_scale)
        if c_fp8 == 2:
            c_scale_out.view(-1)[:M] = so
        Cout[:M] = torch.clamp(v / so[:, None], -448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        return
    Cout[:M] = v.to(Cout.dtype)


def adapter_ln_ok(A, d):
    return REAL.adapter_ln_ok(A, d)


def adapter_ln_fwd(A, R1, R2, Wd, bd, Wu, bu, gamma, beta, eps, act, zp, z, v, y, stats, M=None, y8=None, ys=None, res32=None, y32=None, frag=None):
    """a4r_adapter_ln_fwd: zp = A Wd^T + bd; z = act(zp); v = z Wu^T + bu + R1 + R2; y = LN(v) (bf16 storage points as the kernel's).
    res32: the fp32 twin of the residual operand that is not A; y32: y before its bf16 rounding."""
    M = A.shape[0] if M is None else M
    lo8 = (res32 is not None and res32.dtype == torch.int8) or (y32 is not None and y32.dtype == torch.int8)
    if lo8 and res32 is not None:            # the byte plane joins the bf16 residual that is not A into a 24-bit float
        other = R2 if (R2 is not None and (R1 is A or R1.data_ptr() == A.data_ptr())) else R1
        res32 = (lo4_join if res32.shape[1] == other.shape[1] // 2 else lo8_join)(other[:M], res32[:M])
    if res32 is not None:                    # replaces the residual that is not A
        if R2 is None or R1 is A or R1.data_ptr() == A.data_ptr():
            R1, R2 = (R1, res32) if R2 is not None else (res32, None)
        else:
            R1 = res32
    assert A.dtype == torch.bfloat16 and Wd.shape[0] == 64 and M % 16 == 0
    assert (R1 is A and R2 is not None) or (R2 is A) or (R2 is None and R1 is not A) or \
        (R1.data_ptr() == A.data_ptr() and R2 is not None) or (R2 is not None and R2.data_ptr() == A.data_ptr())
    p = A[:M].float() @ Wd.float().t() + bd
    zp[:M] = p.to(zp.dtype)
    zz = _act(p, act).to(z.dtype)
    z[:M] = zz
    vv = zz.float() @ Wu.float().t() + bu + R1[:M].float() + (R2[:M].float() if R2 is not None else 0)
    vq = vv                                   # (round 4: the LayerNorm runs on the fp32 sum; the bf16 copy v is for the backward)
    if v is not None:
        v[:M] = vv.to(v.dtype)
    mu = vq.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((vq - mu) ** 2).mean(-1, keepdim=True) + eps)
    stats[:M, 0] = mu[:, 0]
    stats[:M, 1] = rstd[:, 0]
    out = (vq - mu) * rstd * gamma + beta
    if y is not None:
        y[:M] = out.to(y.dtype)
    if y32 is not None:
        if lo8:
            y32[:M] = (lo4_of if y32.shape[1] == out.shape[1] // 2 else lo8_of)(out, out.to(torch.bfloat16))
        else:
            y32[:M] = out
    if y8 is not None:
        _quant_rows(out, y8, ys)


def lo4_word_index(H):
    """Position of the nibble word of columns [8 c, 8 c + 8) within a row of the plane (csrc/a4r_adapter_fused.hip: LoWords): the kernel's lane order --
    wave w owns columns [w CW, (w + 1) CW), lane group kg the pieces 32 s + 8 kg + [0, 8); the KS = CW / 32 words of (w, kg) sit together."""
    CW = {128: 32, 256: 32, 512: 64, 768: 96, 1024: 128}[H]
    KS = CW // 32
    col = torch.arange(H // 8) * 8
    w, within = col // CW, col % CW
    s_, kg = within // 32, (within % 32) // 8
    return (w * 4 + kg) * KS + s_


def lo4_of(x, xb):
    """The nibble plane of the 20-bit residual stream (lo4_split8): n = clamp((bits(x) - (bits(bf16) << 16) + 0x800) >> 12, max 7), eight elements per
    32-bit word, element j in bits [4 j, 4 j + 4); returned as int8 [rows, H / 2] (little-endian bytes of those words)."""
    b = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    bf = (xb.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 16
    d = (b - bf) & 0xFFFFFFFF
    d = torch.where(d >= 2 ** 31, d - 2 ** 32, d) + 0x800
    n = (torch.clamp(d, max=0x7FFF) >> 12) & 0xF
    n = n.reshape(n.shape[0], -1, 8)
    words = torch.zeros_like(n)
    words[:, lo4_word_index(x.shape[1])] = n                     # word of columns [8 c, 8 c + 8) -> its place in the kernel's order
    n = words.reshape(n.shape[0], -1, 2)
    byte = n[..., 0] | (n[..., 1] << 4)
    return torch.where(byte >= 128, byte - 256, byte).to(torch.int8)


def lo4_join(xb, lo):
    """bf16 tensor + nibble plane -> the 20-bit float as fp32: (bits(bf16) << 16) + (n << 12)"""
    byte = lo.to(torch.int64) & 0xFF
    n = torch.stack([byte & 0xF, byte >> 4], -1).reshape(lo.shape[0], -1, 8)
    n = n[:, lo4_word_index(xb.shape[1])].reshape(lo.shape[0], -1)          # back to column order
    n = torch.where(n >= 8, n - 16, n)
    bf = (xb.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 16
    bits = (bf + (n << 12)) & 0xFFFFFFFF
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    return bits.view(torch.float32)


def lo8_of(x, xb):
    """The byte plane of the 24-bit residual stream (csrc/a4r_adapter_fused.hip: lo8_split4): the next 8 mantissa bits of the fp32 value x (truncated)
    as a signed offset in [-128, 127] from its bf16 rounding xb -- bit-pattern arithmetic; an exact tie rounded down (+128) is stored as 127."""
    b = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    bf = (xb.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 16
    d = (b - bf) & 0xFFFFFFFF
    d = torch.where(d >= 2 ** 31, d - 2 ** 32, d)
    return (torch.clamp(d, max=0x7FFF) >> 8).to(torch.int8)


def lo8_join(xb, lo):
    """bf16 tensor + byte plane -> the 24-bit float as fp32: (bits(bf16) << 16) + (d << 8)"""
    bf = (xb.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 16
    bits = (bf + (lo.to(torch.int64) << 8)) & 0xFFFFFFFF
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    return bits.view(torch.float32)


def adapter_ln_bwd(dy, v, stats, gamma, dres, zp, act, WuT, WdT, inner_res, dv, dzp, dh, dgamma=None, dbeta=None, dbias=None, M=None,
                   drop_p=0.0, drop_site=0, drop_seed=0, dbd=None, bias_total=False, beta_y=None, frag=None):
    """a4r_adapter_ln_bwd = ln_bwd | (dv Wu) * act'(zp) | dzp Wd (+ dv), with the kernel's bf16 storage points.
    beta_y: `v` is y = LN(v) (the forward did not keep v): xhat = (y - beta_y) / gamma."""
    assert drop_p == 0.0
    M = dy.shape[0] if M is None else M
    if beta_y is not None:
        assert dres is None and dgamma is None and dbeta is None
        xh = (v[:M].float() - beta_y) / gamma
        rstd = stats[:M, 1:2]
        dx = dy[:M].float() * gamma
        g = rstd * (dx - dx.mean(-1, keepdim=True) - xh * (dx * xh).mean(-1, keepdim=True))
        if dbias is not None:
            dbias += g.sum(0)
        dv[:M] = g.to(dv.dtype)
    else:
        ln_bwd(dy, v, stats, gamma, dv, M=M, dgamma=dgamma, dbeta=dbeta, dbias=None if bias_total else dbias, dres=dres)
    dq = dv[:M].float()
    if bias_total and dbias is not None:
        dbias += dq.sum(0)
    dz = (dq @ WuT.float().t()) * _dact(zp[:M].float(), act)
    dzp[:M] = dz.to(dzp.dtype)
    if dbd is not None:
        dbd += dz.sum(0)                 # (the kernel sums the fp32 values, before their bf16 store)
    o = dzp[:M].float() @ WdT.float().t()
    if inner_res:
        o = o + dq
    dh[:M] = o.to(dh.dtype)


def gemm_tn(X, Y, Cacc, M=None):
    M = X.shape[0] if M is None else M
    assert M % 64 == 0 and X.shape[1] % 64 == 0 and Y.shape[1] % 64 == 0
    Cacc += X[:M].float().t() @ Y[:M].float()


def gemm_tn_bias(X, Y, Cacc, xsum, M=None):
    gemm_tn(X, Y, Cacc, M=M)
    colsum(X, xsum, M=M)


def gemm_tn_multi(probs, M=None):
    for X, Y, Cacc, xsum in probs:
        gemm_tn(X, Y, Cacc, M=M)
        if xsum is not None:
            colsum(X, xsum, M=M)


def gemm_tn2(X1, Y1, C1, X2, Y2, C2, M=None, xsum1=None, xsum2=None):
    gemm_tn(X1, Y1, C1, M=M)
    gemm_tn(X2, Y2, C2, M=M)
    m = X1.shape[0] if M is None else M
    if xsum1 is not None:
        xsum1.view(-1)[:X1.shape[1]] += X1[:m].float().sum(0)
    if xsum2 is not None:
        xsum2.view(-1)[:X2.shape[1]] += X2[:m].float().sum(0)


def colsum(X, out, M=None):
    M = X.shape[0] if M is None else M
    out += X[:M].float().sum(0)


FMIN = float(torch.finfo(torch.float32).min)


def _attn(qkv, key_mask, n_items, S, nh, dh, offs, causal, scale, mask_neg):
    Hd = nh * dh
    q, k, v = [qkv[:n_items * S, o:o + Hd].view(n_items, S, nh, dh).transpose(1, 2) for o in offs]
    sc = q @ k.transpose(-1, -2) * scale
    if key_mask is None:
        key_mask = torch.ones(n_items, S)
    allowed = (key_mask != 0)[:, None, None, :].expand(n_items, 1, S, S)
    if causal:
        allowed = torch.tril(allowed)
    sc = sc + torch.where(allowed, torch.tensor(0.0), torch.tensor(mask_neg))
    return (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(n_items * S, Hd)


def _packed_index(offsets, n_items, S):
    """rows of the packed tensors as a padded [n_items, S] index (pad entries -> row 0) + the key mask that goes with it"""
    off = offsets.long().cpu()
    lens = off[1:n_items + 1] - off[:n_items]
    t = torch.arange(S)[None, :]
    valid = t < lens[:, None]
    idx = torch.where(valid, off[:n_items, None] + t, torch.zeros(1, dtype=torch.long))
    return idx, valid


def attn_fwd(qkv, out, key_mask, n_items, S, n_heads, dh, q_off, k_off, v_off, causal, scale, mask_neg,
             drop_p=0.0, drop_site=0, drop_seed=0, offsets=None):
    assert drop_p == 0.0
    if offsets is not None:              # packed items (a4r_attn_t.offsets): unpack to [n_items, S] with the pad keys masked, run, pack back
        idx, valid = _packed_index(offsets, n_items, S)
        o = _attn(qkv.float()[idx.reshape(-1)], valid.float(), n_items, S, n_heads, dh, (q_off, k_off, v_off), causal, scale, mask_neg)
        out[idx[valid], :n_heads * dh] = o.view(n_items, S, -1)[valid].to(out.dtype)
        return
    out[:n_items * S, :n_heads * dh] = _attn(qkv.float(), key_mask, n_items, S, n_heads, dh, (q_off, k_off, v_off), causal, scale, mask_neg).to(out.dtype)


def attn_bwd(qkv, dout, dqkv, key_mask, n_items, S, n_heads, dh, q_off, k_off, v_off, causal, scale, mask_neg,
             drop_p=0.0, drop_site=0, drop_seed=0, offsets=None):
    assert drop_p == 0.0
    if offsets is not None:
        idx, valid = _packed_index(offsets, n_items, S)
        with torch.enable_grad():
            q = qkv.float()[idx.reshape(-1)].clone().requires_grad_(True)
            o = _attn(q, valid.float(), n_items, S, n_heads, dh, (q_off, k_off, v_off), causal, scale, mask_neg)
            do = dout.float()[idx.reshape(-1), :n_heads * dh] * valid.reshape(-1, 1).float()
            o.backward(do)
        g = q.grad.view(n_items, S, -1)[valid]
        for off in (q_off, k_off, v_off):
            dqkv[idx[valid], off:off + n_heads * dh] = g[:, off:off + n_heads * dh].to(dqkv.dtype)
        return
    with torch.enable_grad():
        q = qkv.float().clone().requires_grad_(True)
        o = _attn(q, key_mask, n_items, S, n_heads, dh, (q_off, k_off, v_off), causal, scale, mask_neg)
        o.backward(dout[:n_items * S, :n_heads * dh].float())
    for off in (q_off, k_off, v_off):            # like the kernels: only the columns of the heads are written
        dqkv[:n_items * S, off:off + n_heads * dh] = q.grad[:n_items * S, off:off + n_heads * dh].to(dqkv.dtype)


def attn_long_fwd(qkv, out, lse, n_items, S, n_heads, dh, q_off, k_off, v_off, scale, drop_p=0.0, drop_site=0, drop_seed=0, key_mask=None, causal=False):
    assert dh in (32, 64) and S <= 256 and drop_p == 0.0
    Hd = n_heads * dh
    x = qkv.float()
    q, k = [x[:n_items * S, o:o + Hd].view(n_items, S, n_heads, dh).transpose(1, 2) for o in (q_off, k_off)]
    km = torch.ones(n_items, S) if key_mask is None else key_mask[:n_items].float()
    neg = -1e9 if causal else FMIN
    allowed = (km != 0)[:, None, None, :].expand(n_items, 1, S, S)
    if causal:
        allowed = torch.tril(allowed)
    sc = q @ k.transpose(-1, -2) * scale + torch.where(allowed, torch.tensor(0.0), torch.tensor(neg))
    lse.view(-1)[:n_items * n_heads * S] = torch.logsumexp(sc, -1).reshape(-1)
    out[:n_items * S] = _attn(x, km, n_items, S, n_heads, dh, (q_off, k_off, v_off), causal, scale, neg).to(out.dtype)


def attn_long_bwd(qkv, out, dout, dqkv, lse, delta_ws, n_items, S, n_heads, dh, q_off, k_off, v_off, scale,
                  drop_p=0.0, drop_site=0, drop_seed=0, key_mask=None, causal=False):
    assert drop_p == 0.0
    km = torch.ones(n_items, S) if key_mask is None else key_mask[:n_items].float()
    attn_bwd(qkv, dout, dqkv, km, n_items, S, n_heads, dh, q_off, k_off, v_off, causal, scale, -1e9 if causal else FMIN)


def patchify(img, out, patch, keep_idx=None):
    if img.dtype == torch.uint8:
        img = ((img.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)
    n, C, Hi, Wi = img.shape
    cols = torch.nn.functional.unfold(img.float(), kernel_size=patch, stride=patch).transpose(1, 2)     # [n, NP, C*P*P] in (c, ky, kx) order
    if keep_idx is not None:
        cols = torch.gather(cols, 1, keep_idx.long()[:, :, None].expand(-1, -1, cols.shape[2]))
    rows = cols.reshape(-1, cols.shape[2])
    out[:rows.shape[0], :rows.shape[1]] = rows.to(out.dtype)


def mae_keep_indices(keep, n_patches, noise=None, seed=0, site=0):
    if noise is None:
        noise = torch.rand(keep.shape[0], n_patches, generator=torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF))
    keep[:] = torch.argsort(noise.float(), dim=1, stable=True)[:, :keep.shape[1]].to(torch.int32)


def resample_u8(src, dst, bounds, kk, n_outer, in_len, out_len, inner):
    s = src.reshape(n_outer, in_len, inner).long()
    out = torch.zeros(n_outer, out_len, inner, dtype=torch.long)
    for xx in range(out_len):
        x0, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = torch.full((n_outer, inner), 1 << 21, dtype=torch.long)
        for x in range(cnt):
            acc += s[:, x0 + x] * int(kk[xx, x])
        out[:, xx] = (acc >> 22).clamp(0, 255)
    dst.view(-1)[:] = out.reshape(-1).to(torch.uint8)


def vit_assemble(patches, cls, pos, out, n_items, n_keep, keep_idx=None, tokens_out=0):
    H = cls.numel()
    S_out = tokens_out or n_keep + 1
    x = patches[:n_items * n_keep, :H].float().view(n_items, n_keep, H)
    idx = keep_idx.long() if keep_idx is not None else torch.arange(n_keep)[None].expand(n_items, -1)
    x = x + pos[1:][idx]
    tok = torch.cat([(cls + pos[0])[None, None].expand(n_items, 1, H), x], 1)
    out[:n_items * S_out].view(n_items, S_out, -1)[:, :n_keep + 1] = tok.to(out.dtype)


def _pos_ids(ids, n_items, S, roberta, pad_id):
    idv = ids[:, :S]
    if roberta:
        m = (idv != pad_id).long()
        return idv, torch.cumsum(m, 1) * m + pad_id
    return idv, torch.arange(S).expand(n_items, S)


def embed_bwd(ids, dpre, dword, dpos, n_items, S, roberta=False, pad_id=0):
    idv, pid = _pos_ids(ids, n_items, S, roberta, pad_id)
    g = dpre[:n_items * S].float()
    if dword is not None:
        dword.index_add_(0, idv.reshape(-1), g)
    if dpos is not None:
        dpos.index_add_(0, pid.reshape(-1), g)


def embed_ln(ids, word, pos, type0, gamma, beta, eps, out, n_items, S, roberta=False, pad_id=0,
             drop_p=0.0, drop_site=0, drop_seed=0, pre_out=None, stats_out=None, key_mask_out=None):
    assert drop_p == 0.0
    if key_mask_out is not None:
        key_mask_out[:n_items] = ids[:n_items, S:2 * S].float()
    idv, pid = _pos_ids(ids, n_items, S, roberta, pad_id)
    x = word[idv] + pos[pid] + type0
    if pre_out is not None:
        pre_out[:n_items * S] = x.view(n_items * S, -1).to(pre_out.dtype)
    if stats_out is not None:
        xf = x.view(n_items * S, -1)
        stats_out[:n_items * S, 0] = xf.mean(-1)
        stats_out[:n_items * S, 1] = torch.rsqrt(xf.var(-1, unbiased=False) + eps)
    out[:n_items * S] = torch.nn.functional.layer_norm(x, (x.shape[-1],), gamma, beta, eps).view(n_items * S, -1).to(out.dtype)


def _quant_rows(x, q, scale):
    amax = x.abs().amax(1)
    inv = torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
    q[:x.shape[0]] = (x * inv[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    scale.view(-1)[:x.shape[0]] = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))


def quant_rows_fp8(x, q, scale, M=None):
    M = x.shape[0] if M is None else M
    _quant_rows(x[:M].float(), q, scale)


def ln_fwd_sum(h, res, gamma, beta, eps, y, stats, M=None, res32=None, sum_out=None, sum32=None, y32=None):
    M = h.shape[0] if M is None else M
    x = h[:M].float() + (res32[:M] if res32 is not None else res[:M].float())
    if sum_out is not None:
        sum_out[:M] = x.to(sum_out.dtype)
    if sum32 is not None:
        sum32[:M] = x
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
    stats[:M, 0] = mu[:, 0]
    stats[:M, 1] = rstd[:, 0]
    out = (x - mu) * rstd * gamma + beta
    y[:M] = out.to(y.dtype)
    if y32 is not None:
        y32[:M] = out


def ln_fwd(v, gamma, beta, eps, y, stats, M=None, add=None, drop_p=0.0, drop_site=0, drop_seed=0, y8=None, ys=None):
    assert drop_p == 0.0
    M = v.shape[0] if M is None else M
    x = v[:M].float()
    if add is not None:
        x = x + add[torch.arange(M) % add.shape[0]]
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    if stats is not None:
        stats[:M, 0] = mu[:, 0]
        stats[:M, 1] = rstd[:, 0]
    out = (x - mu) * rstd * gamma + beta
    if y is not None:
        y[:M] = out.to(y.dtype)
    if y8 is not None:
        _quant_rows(out, y8, ys)


def ln_bwd(dy, v, stats, gamma, dv, M=None, add=None, dgamma=None, dbeta=None, dbias=None, dres=None,
           drop_p=0.0, drop_site=0, drop_seed=0, dv2=None, drop2_p=0.0, drop2_site=0, drop2_seed=0):
    assert drop_p == 0.0 and drop2_p == 0.0
    M = v.shape[0] if M is None else M
    x = v[:M].float()
    if add is not None:
        x = x + add[torch.arange(M) % add.shape[0]]
    mu, rstd = stats[:M, 0:1], stats[:M, 1:2]
    xh = (x - mu) * rstd
    d = dy[:M].float()
    if dgamma is not None:
        dgamma += (d * xh).sum(0)
    if dbeta is not None:
        dbeta += d.sum(0)
    dx = d * gamma
    g = rstd * (dx - dx.mean(-1, keepdim=True) - xh * (dx * xh).mean(-1, keepdim=True))
    if dbias is not None:
        dbias += g.sum(0)
    if dres is not None:
        g = g + dres[:M].float()
    dv[:M] = g.to(dv.dtype)
    if dv2 is not None:
        dv2[:M] = g.to(dv2.dtype)


def _f32_at(ptr, n):
    """fp32 view of host memory at a raw address (the descriptor tables carry data_ptr()s of CPU tensors here)."""
    import numpy as np
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr)))


def _parse(desc_dev, cls, n):
    raw = bytes(desc_dev.numpy().tobytes())
    return [cls.from_buffer_copy(raw[i * ctypes.sizeof(cls):(i + 1) * ctypes.sizeof(cls)]) for i in range(n)]


def scatter_rows_fill(src, dst, n, row_step, fill_rows):
    dst[:fill_rows] = 0
    dst[0:n * row_step:row_step] = src[:n]


def zero(t):
    t.zero_()


def lora_merge(W, A, B, scaling, dst, dstT, r):
    w = W.float() + (scaling * (B.float() @ A.float()) if r else 0)
    dst.copy_(w.to(dst.dtype))
    dstT.copy_(w.t().to(dstT.dtype))


def lora_table(entries, device):
    return list(entries)


def lora_merge_batch(tab):
    for W, A, B, s, dst, dstT, r in tab:
        lora_merge(W, A, B, s, dst, dstT, r)


def lora_bwd_fused_ok(x, M, H):
    return x.dtype == torch.bfloat16 and H == 768 and M % 16 == 0


def lora_bwd_fused(x, dqa, dqb, Aa, Ab, BTa, BTb, scale_a, scale_b, dAa, dAb, dBa, dBb, dbias_a, dbias_b, M, rank_rows=8):
    """include/a4r.h: t and dt rounded to the element type between the two stages"""
    R = rank_rows
    nb = R - (R == 16)                                                     # dB columns written (rank 15 of the wide form is the row of ones)
    xf, qa, qb = x[:M].float(), dqa[:M].float(), dqb[:M].float()
    A = torch.cat([Aa[:R].float(), Ab[:R].float()], 0)                     # [2 R, H]
    t = (xf @ A.t()).to(x.dtype).float()                                   # [M, 2 R]
    dt = torch.cat([(qa @ BTa[:R].float().t()) * scale_a, (qb @ BTb[:R].float().t()) * scale_b], 1).to(x.dtype).float()
    dA = dt.t() @ xf                                                       # [2 R, H]
    dAa[:R] += dA[:R]
    dAb[:R] += dA[R:]
    dBa[:, :nb] += qa.t() @ t[:, :nb]
    dBb[:, :nb] += qb.t() @ t[:, R:R + nb]
    if dbias_a is not None:
        dbias_a += qa.sum(0)
    if dbias_b is not None:
        dbias_b += qb.sum(0)


def _phm_E(params, d):
    n, ip, oq = d.n, d.in_f // d.n, d.out_f // d.n
    rule = params[d.rule_off:d.rule_off + n ** 3].view(n, n, n)
    wl = params[d.wl_off:d.wl_off + n * ip].view(n, ip)
    wr = params[d.wr_off:d.wr_off + n * oq].view(n, oq)
    return rule, wl, wr


def phm_build(params, desc_dev, n_desc, eff):
    for d in _parse(desc_dev, PhmDesc, n_desc):
        rule, wl, wr = _phm_E(params, d)
        E = torch.einsum('kab,kp,kq->bqap', rule, wl, wr).reshape(d.out_f, d.in_f)         # E[b oq + q][a ip + p]
        eff[d.out_off:d.out_off + d.in_f * d.out_f] = E.reshape(-1)


def phm_bwd(params, desc_dev, n_desc, grads):
    for d in _parse(desc_dev, PhmDesc, n_desc):
        n, ip, oq = d.n, d.in_f // d.n, d.out_f // d.n
        rule, wl, wr = _phm_E(params, d)
        G = _f32_at(d.G, (d.out_f - 1) * d.ldg + d.in_f)
        G = torch.as_strided(G, (d.out_f, d.in_f), (d.ldg, 1)).reshape(n, oq, n, ip)      # [b, q, a, p]
        grads[d.rule_off:d.rule_off + n ** 3] += torch.einsum('bqap,kp,kq->kab', G, wl, wr).reshape(-1)
        grads[d.wl_off:d.wl_off + n * ip] += torch.einsum('bqap,kab,kq->kp', G, rule, wr).reshape(-1)
        grads[d.wr_off:d.wr_off + n * oq] += torch.einsum('bqap,kab,kp->kq', G, rule, wl).reshape(-1)


def unpack_add(target, desc_dev, n_desc, max_elems):
    for d in _parse(desc_dev, AddDesc, n_desc):
        src = torch.as_strided(_f32_at(d.src, (d.rows - 1) * d.ld + d.cols), (d.rows, d.cols), (d.ld, 1))
        target.view(-1)[d.dst_off:d.dst_off + d.rows * d.cols] += d.alpha * src.reshape(-1)


def dropout_apply(x, y, drop_p, drop_site, drop_seed, M=None):
    raise AssertionError('dropout is off in the simulator')


def gather_rows(src, dst, n, row_step):
    dst[:n] = src[0:n * row_step:row_step]


def rows_idx_copy(src, dst, idx, n, scatter=False):
    if scatter:
        dst[idx[:n].long()] = src[:n]
    else:
        dst[:n] = src[idx[:n].long()]


def scatter_rows(src, dst, n, row_step):
    dst[0:n * row_step:row_step] = src[:n]


def act_bwd_f32(dy, pre, dx, act):
    dx.copy_(dy * _dact(pre, act))


def _valid(log_mask, B, T, cpc):
    if cpc:
        m = torch.zeros(B, T, dtype=torch.bool)
        m[:, -1] = True
        return m
    return log_mask.view(B, T) != 0


def score_bce_fwd(emb, prec, log_mask, pos, neg, loss_ws, B, L, E, cpc):
    T = L - 1
    e = emb[:B * L * 2].view(B, L, 2, E)
    p = prec[:B * T].view(B, T, E)
    ps, ns = (p * e[:, 1:, 0]).sum(-1), (p * e[:, :-1, 1]).sum(-1)
    pos.view(-1)[:B * T] = ps.reshape(-1)
    neg.view(-1)[:B * T] = ns.reshape(-1)
    m = _valid(log_mask, B, T, cpc)
    sp = torch.nn.functional.softplus
    s = (sp(-ps[m]) + sp(ns[m])).sum()
    loss_ws.view(-1)[1] = s
    loss_ws.view(-1)[2] = float(m.sum())
    loss_ws.view(-1)[0] = s / m.sum()


def score_bce_bwd(emb, prec, log_mask, pos, neg, loss_ws, loss_scale, d_prec, d_emb, B, L, E, cpc, scale_dev=None):
    if scale_dev is not None:
        loss_scale = loss_scale * float(scale_dev)
    T = L - 1
    with torch.enable_grad():
        e = emb[:B * L * 2].view(B, L, 2, E).clone().requires_grad_(True)
        p = prec[:B * T].view(B, T, E).clone().requires_grad_(True)
        ps, ns = (p * e[:, 1:, 0]).sum(-1), (p * e[:, :-1, 1]).sum(-1)
        m = _valid(log_mask, B, T, cpc)
        sp = torch.nn.functional.softplus
        loss = (sp(-ps[m]) + sp(ns[m])).sum() / m.sum() * loss_scale
        loss.backward()
    d_prec[:B * T] = p.grad.view(B * T, E)
    d_emb[:B * L * 2] = e.grad.view(B * L * 2, E)


def emb_grad_add_inputs(d_in, d_emb, B, L, E):
    T = L - 1
    d_emb[:B * L * 2].view(B, L, 2, E)[:, :-1, 0] += d_in[:B * T].view(B, T, E)


def take_inputs(emb, out, B, L, E):
    T = L - 1
    out[:B * T] = emb[:B * L * 2].view(B, L, 2, E)[:, :-1, 0].reshape(B * T, E)


def adam_step(p, g, m, v, seg_end, seg_group, group_lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    idx = torch.arange(p.numel())
    seg = torch.searchsorted(seg_end.long(), idx, right=True).clamp(max=seg_end.numel() - 1)
    lr = group_lr[seg_group.long()[seg]]
    gi = g * grad_scale
    m.mul_(beta1).add_(gi, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps))


def pack_matrices(flat, desc_dev, n_desc, max_elems, dtype):
    raw = bytes(desc_dev.numpy().tobytes())
    arr = (PackDesc * n_desc).from_buffer_copy(raw)
    tdt = torch.bfloat16 if dtype == BF16 else torch.float32
    for d in arr:
        src = flat[d.src_off:d.src_off + d.rows * d.cols].view(d.rows, d.cols)
        if d.transpose & 1:
            src = src.t()
        layout = d.transpose >> 1
        ld = d.cols_pad if layout else (d.dst_ld or d.cols_pad)
        n = (d.rows_pad - 1) * ld + d.cols_pad
        buf = (ctypes.c_char * (n * (2 if dtype == BF16 else 4))).from_address(d.dst)
        dst = torch.frombuffer(buf, dtype=tdt).as_strided((d.rows_pad, d.cols_pad), (ld, 1))
        if layout:           # fragment order of the one-launch adapter kernels (csrc/a4r_head.hip: pack_dst_index)
            full = torch.zeros(d.rows_pad, d.cols_pad, dtype=tdt)
            full[:src.shape[0], :src.shape[1]] = src.to(tdt)
            dst.view(-1)[frag_index(layout, d.rows_pad, d.cols_pad).view(-1)] = full.view(-1)
            continue
        dst.zero_()
        dst[:src.shape[0], :src.shape[1]] = src.to(tdt)


def frag_index(layout, rows_pad, cols_pad):
    """destination index of element (r, c) for a4r_pack_desc_t layouts 1 ([64, H]) and 2 ([H, 64]) -- restated from include/a4r.h"""
    H = cols_pad if layout == 1 else rows_pad
    NW = 4 if H == 128 else 8
    CW = H // NW
    KS = CW // 32
    r = torch.arange(rows_pad).view(-1, 1).expand(rows_pad, cols_pad)
    c = torch.arange(cols_pad).view(1, -1).expand(rows_pad, cols_pad)
    if layout == 1:
        w, cc = c // CW, c % CW
        s_, kg, j, nt, fr = cc // 32, (cc % 32) // 8, cc % 8, r // 16, r % 16
        return ((((w * KS + s_) * 4 + nt) * 64 + kg * 16 + fr) * 8 + j)
    w, rr = r // CW, r % CW
    s_, q = rr // 32, rr % 32
    fr, h, ks, kg, j = (q // 8) * 4 + (q % 4), (q // 4) % 2, c // 32, (c % 32) // 8, c % 8
    return (((((w * KS + s_) * 2 + h) * 2 + ks) * 64 + kg * 16 + fr) * 8 + j)


def eval_rank(prec, item_emb, target, hist_ptr, hist_idx, rank):
    sc = prec @ item_emb.t()
    for u in range(prec.shape[0]):
        s = sc[u].clone()
        ts = s[int(target[u])].item()
        h = hist_idx[int(hist_ptr[u]):int(hist_ptr[u + 1])].long()
        s[h] = -float('inf')
        rank[u] = int((s[1:] > ts).sum()) + 1


# ------------------------------------------------------------------ a4r_sasrec_block_fwd / _bwd (one launch per SASRec block)
def _sasrec_block_fn(d, x, log_mask, T):
    """The block's forward as differentiable torch ops (dropout off: the CPU suite has none).  x [B * T, 64]."""
    B = x.shape[0] // T
    E, nh, dh = 64, 2, 32
    xx = x.view(B, T, E)
    qkv = xx @ d['wqkv'].t()
    q, k, v = (t.view(B, T, nh, dh).transpose(1, 2) for t in qkv.split(E, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    allowed = (log_mask[:, None, None, :] != 0) & torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    s = torch.where(allowed, s, s + d['mask_neg'])
    ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, E)
    h = ctx @ d['wfc'].t()

    def adapter(hh, k_):
        dd = d['d']
        zp = hh @ d['wd' + k_][:dd].t() + d['bd' + k_][:dd]
        return _act(zp, d['act']) @ d['wu' + k_][:, :dd].t() + d['bu' + k_] + (hh if d['inner_res'] else 0)

    def ln(vv, g, b):
        mu = vv.mean(-1, keepdim=True)
        return (vv - mu) * torch.rsqrt(((vv - mu) ** 2).mean(-1, keepdim=True) + d['eps']) * g + b
    if d.get('mode', 0) == 1:                      # SASRecPfeifferAdaptedSelfOutput (model.py:458-471)
        x1 = ln(xx + h, d['ln1_g'], d['ln1_b'])
        va = torch.relu(x1 @ d['w1'].t() + d['b1']) @ d['w2'].t() + d['b2'] + x1
        t = ln(va, d['ln2_g'], d['ln2_b'])
        return ln(adapter(t, '2') + va, d['ln3_g'], d['ln3_b']).reshape(B * T, E)
    x1 = ln(xx + adapter(h, '1'), d['ln1_g'], d['ln1_b'])
    h2 = torch.relu(x1 @ d['w1'].t() + d['b1']) @ d['w2'].t() + d['b2']
    return ln(x1 + adapter(h2, '2'), d['ln2_g'], d['ln2_b']).reshape(B * T, E)


def sasrec_block(desc, x, log_mask, out, n_users, T, train, dy=None):
    assert not (train and (desc['drop_attn'] or desc['drop_hidden'])), 'sim_lib has no dropout'
    n = n_users * T
    if dy is None:
        with torch.no_grad():
            out[:n] = _sasrec_block_fn(desc, x[:n], log_mask[:n_users], T)
        return
    names = ('wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2') + (('ln3_g', 'ln3_b') if desc.get('mode', 0) == 1 else ())
    with torch.enable_grad():                       # (called from inside an autograd Function's backward: grad mode is off there)
        leaf = {k: desc[k].detach().clone().requires_grad_(True) for k in names}
        xin = x[:n].detach().clone().requires_grad_(True)
        y = _sasrec_block_fn(dict(desc, **leaf), xin, log_mask[:n_users], T)
        grads = torch.autograd.grad(y, [xin] + [leaf[k] for k in names], dy[:n], allow_unused=True)
    out[:n] = grads[0]
    dd = desc['d']
    for k, g in zip(names, grads[1:]):
        g = g.detach() if g is not None else None
        tgt = desc.get('g_' + k)
        if tgt is None or g is None:
            continue
        if k.startswith('wd'):
            tgt[:dd, :64] += g[:dd]
        elif k.startswith('wu'):
            tgt[:64, :dd] += g[:, :dd]
        elif k.startswith('bd'):
            tgt[:dd] += g[:dd]
        else:
            tgt += g
