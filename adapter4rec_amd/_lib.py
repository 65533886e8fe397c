"""ctypes binding of liba4r_hip.so (C ABI declared in include/a4r.h).

The library is the product's only compute path: if it is missing or a call fails this module
raises -- there is no PyTorch / CPU fallback.  Tensors are passed as raw device pointers plus
leading dimensions; kernels are enqueued on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('A4R_LIB_PATH') or os.path.join(_HERE, 'liba4r_hip.so')    # A4R_LIB_PATH: A/B builds (tools/), same C ABI

ABI_VERSION = 409          # = A4R_ABI_VERSION of include/a4r.h (tests/test_abi_cpu.py compares the two)
BF16, F32, FP8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_TANH, ACT_LEAKY = 0, 1, 2, 3, 4
DACT_MUL = 15
DACT_MUL_Q8 = 14          # Pre = the uint8 derivative tensor a c2_deriv='q8' launch wrote (include/a4r.h: c2_mode 2)
Q8_OFF, Q8_STEP = 0.1289, 0.0049326
EVAL_MAX_HISTORY = 264         # A4R_EVAL_MAX_HISTORY (include/a4r.h)
ACT_BY_NAME = {'none': 0, 'relu': 1, 'RELU': 1, 'gelu': 2, 'GELU': 2, 'gelu_new': 3, 'leaky_relu': 4}

EXPORTS = [
    'a4r_version', 'a4r_gemm_nt', 'a4r_gemm_tn', 'a4r_gemm_tn_bias', 'a4r_gemm_tn_multi', 'a4r_gemm_tn2', 'a4r_colsum', 'a4r_attn_fwd', 'a4r_attn_bwd', 'a4r_embed_ln',
    'a4r_ln_fwd', 'a4r_ln_bwd', 'a4r_gather_rows', 'a4r_scatter_rows', 'a4r_rows_idx_copy', 'a4r_act_bwd_f32', 'a4r_score_bce_fwd',
    'a4r_score_bce_bwd', 'a4r_emb_grad_add_inputs', 'a4r_take_inputs', 'a4r_adam_step', 'a4r_pack_matrices',
    'a4r_eval_rank', 'a4r_dropout_apply', 'a4r_gemm_variant', 'a4r_gemm_tail_plan', 'a4r_gemm_tail_max', 'a4r_gemm_rows_256', 'a4r_adapter_ln_fwd', 'a4r_adapter_ln_bwd', 'a4r_ln_fwd_fp8', 'a4r_ln_fwd_sum', 'a4r_quant_rows_fp8', 'a4r_lora_merge', 'a4r_lora_merge_batch', 'a4r_lora_bwd_fused', 'a4r_lora_bwd_fused_ws_floats', 'a4r_phm_build', 'a4r_phm_bwd', 'a4r_unpack_add', 'a4r_memset_zero',
    'a4r_sasrec_block_fwd', 'a4r_sasrec_block_bwd', 'a4r_scatter_rows_fill', 'a4r_attn_long_fwd', 'a4r_attn_long_bwd', 'a4r_patchify', 'a4r_vit_assemble', 'a4r_resample_u8', 'a4r_embed_bwd', 'a4r_mae_keep_indices',
    'a4r_encoder_layer_fwd', 'a4r_encoder_layer_bwd',
]


class GemmArgs(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p), ('C2', C.c_void_p),
                ('R1', C.c_void_p), ('R2', C.c_void_p), ('Pre', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('lda', C.c_int32), ('ldb', C.c_int32), ('ldc', C.c_int32), ('ldc2', C.c_int32),
                ('ldr1', C.c_int32), ('ldr2', C.c_int32), ('ldpre', C.c_int32),
                ('in_dtype', C.c_int32), ('out_dtype', C.c_int32), ('act', C.c_int32), ('dact', C.c_int32),
                ('drop_first', C.c_int32), ('c2_mode', C.c_int32), ('alpha', C.c_float), ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('drop_row0', C.c_int64), ('scale_a', C.c_void_p), ('scale_b', C.c_void_p),
                ('c_fp8', C.c_int32), ('c_scale', C.c_float), ('c_scale_out', C.c_void_p), ('q8_tiled', C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [('qkv', C.c_void_p), ('ld', C.c_int32), ('q_off', C.c_int32), ('k_off', C.c_int32), ('v_off', C.c_int32),
                ('out', C.c_void_p), ('ldo', C.c_int32), ('dout', C.c_void_p), ('dqkv', C.c_void_p), ('key_mask', C.c_void_p),
                ('n_items', C.c_int32), ('S', C.c_int32), ('n_heads', C.c_int32), ('dh', C.c_int32), ('causal', C.c_int32),
                ('dtype', C.c_int32), ('scale', C.c_float), ('mask_neg', C.c_float),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64), ('offsets', C.c_void_p)]


class PackDesc(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst', C.c_void_p), ('rows', C.c_int32), ('cols', C.c_int32),
                ('rows_pad', C.c_int32), ('cols_pad', C.c_int32), ('transpose', C.c_int32), ('dst_ld', C.c_int32)]


class PhmDesc(C.Structure):
    _fields_ = [('rule_off', C.c_int64), ('wl_off', C.c_int64), ('wr_off', C.c_int64), ('out_off', C.c_int64), ('G', C.c_void_p),
                ('ldg', C.c_int32), ('in_f', C.c_int32), ('out_f', C.c_int32), ('n', C.c_int32), ('pad_', C.c_int32)]


class LayerAdapter(C.Structure):
    """a4r_layer_adapter_t (include/a4r.h)."""
    _fields_ = [(n, C.c_void_p) for n in ('wd', 'wu', 'wdT', 'wuT', 'wd_f', 'wu_f', 'wdT_f', 'wuT_f', 'bd', 'bu', 'g_wu', 'g_wd', 'g_bu', 'g_bd')] + \
               [(n, C.c_int32) for n in ('ldg_wu', 'ldg_wd', 'act', 'pad_')]


class EncoderLayer(C.Structure):
    """a4r_encoder_layer_t (include/a4r.h): one post-LN encoder layer with serial Houlsby adapters for a4r_encoder_layer_fwd / _bwd."""
    _fields_ = [(n, C.c_int32) for n in ('M', 'H', 'F', 'n_items', 'S', 'n_heads', 'dh', 'causal')] + \
               [(n, C.c_float) for n in ('scale', 'mask_neg', 'ln_eps', 'p_attn', 'p_hidden')] + [('drop_site', C.c_uint32), ('drop_seed', C.c_uint64)] + \
               [(n, C.c_void_p) for n in ('key_mask', 'offsets', 'wqkv', 'wqkvT', 'wo', 'woT', 'wi', 'wiT', 'wo2', 'wo2T',
                                          'bqkv', 'bo', 'bi', 'bo2', 'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b')] + \
               [('ad', LayerAdapter * 2)] + \
               [(n, C.c_void_p) for n in ('qkv', 'ctx', 'h1', 'v1', 'zp1', 'z1', 'u', 'upre', 'h2', 'v2', 'zp2', 'z2', 'st1', 'st2')] + \
               [('upre_q8', C.c_int32), ('q8_tiled', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('x_lo', 'x1_lo', 'xout_lo', 'dv1', 'dv2', 'dzp', 'd_h', 'du', 'dx1', 'dctx', 'dqkv')] + [('lo_nibble', C.c_int32)]


class SasrecBlock(C.Structure):
    """a4r_sasrec_block_t (include/a4r.h)."""
    _PTRS = ('wqkv', 'wfc', 'w1', 'b1', 'w2', 'b2', 'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b', 'wd1', 'bd1', 'wu1', 'bu1', 'wd2', 'bd2', 'wu2', 'bu2',
             'g_wd1', 'g_bd1', 'g_wu1', 'g_bu1', 'g_wd2', 'g_bd2', 'g_wu2', 'g_bu2')
    _fields_ = [(n, C.c_void_p) for n in _PTRS] + \
               [(n, C.c_int32) for n in ('E', 'n_heads', 'F', 'd', 'ldwu', 'ldg_d', 'ldg_u', 'act', 'inner_res')] + \
               [(n, C.c_float) for n in ('eps', 'mask_neg', 'drop_attn', 'drop_hidden')] + [('drop_site', C.c_uint32), ('drop_seed', C.c_uint64)] + \
               [('mode', C.c_int32)] + [(n, C.c_void_p) for n in ('ln3_g', 'ln3_b', 'g_ln3_g', 'g_ln3_b')]


class AddDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst_off', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32), ('ld', C.c_int32), ('alpha', C.c_float)]


_lib = None


def lib():
    """Load the HIP library once; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  adapter4rec_amd has no CPU / PyTorch fallback.')
        _lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(_lib, name).restype = C.c_int
        got = _lib.a4r_version()
        if got != ABI_VERSION:        # an older A/B build has every export but other argument lists: calling it would pass shifted pointers
            _lib = None
            raise RuntimeError(f'{LIB_PATH} has ABI version {got}, this binding is for {ABI_VERSION}: rebuild it (make -C adapter4rec_amd/csrc)')
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed with status {rc} ' + {-1: '(invalid argument)', -2: '(launch failure)'}.get(rc, ''))


_DEV_INDEX = None


def _stream():
    """torch's CURRENT stream on this process's device as a raw hipStream_t.  One process drives one GPU, so the device index is
    resolved once; torch._C._cuda_getCurrentRawStream is the accessor torch's own extensions use (torch.cuda.current_stream()
    builds a Stream object per call: 8 us x 400 launches per step)."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(_DEV_INDEX))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _pi(t):
    return t.data_ptr() if t is not None else 0


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.uint8:           # e4m3 bit patterns (a4r_quant_rows_fp8 / a4r_ln_fwd_fp8 / quantize_weight_fp8)
        return FP8
    raise TypeError(f'unsupported dtype {t.dtype}')


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'row-major 2-D tensor expected'
    return t.stride(0)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('adapter4rec_amd kernels need device tensors (no CPU fallback)')


# ------------------------------------------------------------------ wrappers
def gemm_nt(A, B, Cout, bias=None, C2=None, R1=None, R2=None, Pre=None, act=0, dact=0, alpha=1.0,
            drop_p=0.0, drop_site=0, drop_seed=0, M=None, drop_first=False, c2_deriv=False, scale_a=None, scale_b=None,
            c_fp8=0, c_scale=1.0, c_scale_out=None, q8_tiled=False):
    """q8_tiled: the 8-bit derivative tensor (C2 with c2_deriv='q8' / Pre with DACT_MUL_Q8) in the 256-tile kernel's own order (include/a4r.h).
    c_fp8 (fp8 operands only): Cout is a uint8 tensor that receives the result as e4m3 bytes (include/a4r.h: 1 = static scale c_scale,
    2 = row m scaled by scale_a[m] * c_scale, that product written to c_scale_out[m])."""
    if not (A.is_cuda and B.is_cuda and Cout.is_cuda):
        require_gpu(A, B, Cout)
    N, K = B.shape
    din, dout = _dt(A), _dt(Cout)
    if c_fp8:
        assert din == FP8 and Cout.dtype == torch.uint8 and R1 is None and R2 is None and (c_fp8 == 1 or c_scale_out is not None)
        dout = BF16
    assert A.shape[1] == K and Cout.shape[1] == N and _dt(B) == din
    q8c, q8p = c2_deriv == 'q8', dact == DACT_MUL_Q8
    for t, q8 in ((C2, q8c), (R1, False), (R2, False), (Pre, q8p)):
        assert t is None or (t.dtype == torch.uint8 if q8 else _dt(t) == dout)
    assert bias is None or bias.dtype == torch.float32
    # positional construction (field order of a4r_gemm_t): one C call instead of ~30 attribute stores -- this wrapper runs
    # ~220 times per training step
    g = GemmArgs(A.data_ptr(), B.data_ptr(), Cout.data_ptr(), _pi(bias), _pi(C2), _pi(R1), _pi(R2), _pi(Pre),
                 A.shape[0] if M is None else M, N, K, _ld(A), _ld(B), _ld(Cout),
                 _ld(C2) if C2 is not None else 0, _ld(R1) if R1 is not None else 0, _ld(R2) if R2 is not None else 0,
                 _ld(Pre) if Pre is not None else 0, din, dout, act, dact, int(drop_first), 2 if q8c else int(bool(c2_deriv)), alpha, drop_p, drop_site,
                 drop_seed, 0, _pi(scale_a), _pi(scale_b), int(c_fp8), float(c_scale), _pi(c_scale_out), int(bool(q8_tiled)))
    _check(lib().a4r_gemm_nt(_stream(), C.byref(g)), 'a4r_gemm_nt')


def gemm_variant(v):
    return lib().a4r_gemm_variant(C.c_int(v))


def gemm_tail_max(k):
    return lib().a4r_gemm_tail_max(C.c_int(k))


def gemm_tail_plan(M, N):
    """(p_full, kp): row panels of full 256-row tiles and the height / 32 of the short tiles behind them (kp = 0: none)."""
    pf, kp = C.c_int(0), C.c_int(0)
    lib().a4r_gemm_tail_plan(C.c_int(M), C.c_int(N), C.byref(pf), C.byref(kp))
    return pf.value, kp.value


def gemm_rows_256(M, N):
    """Leading rows of an [M, N] gemm_nt output that the 256-tile kernel computes (= the tile-native part of a q8_tiled tensor)."""
    return lib().a4r_gemm_rows_256(C.c_int(M), C.c_int(N))


def adapter_ln_ok(A, d):
    """Shapes the one-launch adapter kernels (a4r_adapter_ln_fwd / _bwd) are instantiated for."""
    return A.dtype == torch.bfloat16 and d == 64 and A.shape[1] in (128, 256, 512, 768, 1024)


def adapter_ln_fwd(A, R1, R2, Wd, bd, Wu, bu, gamma, beta, eps, act, zp, z, v, y, stats, M=None, y8=None, ys=None, res32=None, y32=None, frag=None):
    """res32 / y32 (fp32 [M, H], optional): the residual operand that is not A read in fp32, and y before its bf16 rounding (include/a4r.h).
    The same two as int8 [M, H] (--residual_dtype bf24, w_frag bit 1): byte planes beside the bf16 tensors -- residual read / y written as 24-bit floats;
    as int8 [M, H / 2] (--residual_dtype bf20, w_frag bits 1 + 2): nibble planes, 20-bit floats.
    frag = (Wd_f, Wu_f): the same matrices in fragment order (a4r_pack_matrices layouts 1 / 2): read instead of Wd / Wu (w_frag)."""
    require_gpu(A, R1, R2, v, y, res32, y32)
    M = A.shape[0] if M is None else M
    assert y is not None or y8 is not None
    twins = [t for t in (res32, y32) if t is not None]
    lo8 = bool(twins) and twins[0].dtype == torch.int8
    assert all(t.dtype == (torch.int8 if lo8 else torch.float32) and t.shape[0] >= M for t in twins)
    lo4 = lo8 and twins[0].shape[1] == A.shape[1] // 2
    assert not lo8 or all(t.shape[1] == (A.shape[1] // 2 if lo4 else A.shape[1]) for t in twins)
    wd_, wu_ = (Wd, Wu) if frag is None else frag
    _check(lib().a4r_adapter_ln_fwd(_stream(), _p(A), C.c_int(_ld(A)), _p(R1), C.c_int(_ld(R1)), _p(R2), C.c_int(_ld(R2) if R2 is not None else 0),
                                    _p(wd_), _p(bd), _p(wu_), _p(bu), _p(gamma), _p(beta), C.c_float(eps), C.c_int(act),
                                    _p(zp), _p(z), _p(v), C.c_int(_ld(v) if v is not None else 0), _p(y), C.c_int(_ld(y) if y is not None else 0), _p(stats),
                                    C.c_int(M), C.c_int(A.shape[1]), C.c_int(Wd.shape[0]), C.c_int(_dt(A)),
                                    _p(y8), C.c_int(_ld(y8) if y8 is not None else 0), _p(ys),
                                    _p(res32), C.c_int(_ld(res32) if res32 is not None else 0), _p(y32), C.c_int(_ld(y32) if y32 is not None else 0),
                                    C.c_int((0 if frag is None else 1) | (2 if lo8 else 0) | (4 if lo4 else 0))),
           'a4r_adapter_ln_fwd')


def adapter_ln_bwd(dy, v, stats, gamma, dres, zp, act, WuT, WdT, inner_res, dv, dzp, dh, dgamma=None, dbeta=None, dbias=None, M=None,
                   drop_p=0.0, drop_site=0, drop_seed=0, dbd=None, bias_total=False, beta_y=None, frag=None):
    """beta_y: the forward kept y = LN(v) instead of v (called with v=None); `v` is that y and xhat is rebuilt as (y - beta_y) / gamma.
    frag = (WuT_f, WdT_f): the two matrices in fragment order (flags bit 1)."""
    require_gpu(dy, v, dv, dh)
    M = dy.shape[0] if M is None else M
    wut_, wdt_ = (WuT, WdT) if frag is None else frag
    _check(lib().a4r_adapter_ln_bwd(_stream(), _p(dy), C.c_int(_ld(dy)), _p(v), C.c_int(_ld(v)), _p(stats), _p(gamma),
                                    _p(dres), C.c_int(_ld(dres) if dres is not None else 0), _p(zp), C.c_int(act), _p(wut_), _p(wdt_),
                                    C.c_int(int(inner_res)), _p(dv), C.c_int(_ld(dv)), _p(dzp), _p(dh), C.c_int(_ld(dh)),
                                    _p(dgamma), _p(dbeta), _p(dbias), C.c_int(M), C.c_int(dy.shape[1]), C.c_int(WuT.shape[0]), C.c_int(_dt(dy)),
                                    C.c_float(drop_p), C.c_uint32(drop_site), C.c_uint64(drop_seed), _p(dbd),
                                    C.c_int(int(bias_total) | (0 if frag is None else 2)), _p(beta_y)), 'a4r_adapter_ln_bwd')


def sasrec_block(desc, x, log_mask, out, n_users, T, train, dy=None):
    """a4r_sasrec_block_fwd (dy None: out = y) / a4r_sasrec_block_bwd (out = dx; the adapter gradients are added into desc.g_*).
    desc: dict of the a4r_sasrec_block_t fields (tensors for the pointer fields, None = null)."""
    require_gpu(x, log_mask, out, dy)
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and log_mask.dtype == torch.float32 and x.shape[1] == 64 and x.is_contiguous() and out.is_contiguous()
    b = SasrecBlock()
    for k, v in desc.items():
        setattr(b, k, (v.data_ptr() if v is not None else None) if (k in SasrecBlock._PTRS or k in ('ln3_g', 'ln3_b', 'g_ln3_g', 'g_ln3_b')) else v)
    if dy is None:
        _check(lib().a4r_sasrec_block_fwd(_stream(), C.byref(b), _p(x), _p(log_mask), _p(out), C.c_int(n_users), C.c_int(T), C.c_int(int(train))), 'a4r_sasrec_block_fwd')
    else:
        assert dy.dtype == torch.float32 and dy.is_contiguous()
        _check(lib().a4r_sasrec_block_bwd(_stream(), C.byref(b), _p(x), _p(log_mask), _p(dy), _p(out), C.c_int(n_users), C.c_int(T), C.c_int(int(train))),
               'a4r_sasrec_block_bwd')


def gemm_tn(X, Y, Cacc, M=None):
    require_gpu(X, Y, Cacc)
    assert Cacc.dtype == torch.float32 and _dt(X) == _dt(Y)
    M = X.shape[0] if M is None else M
    _check(lib().a4r_gemm_tn(_stream(), _p(X), C.c_int(_ld(X)), _p(Y), C.c_int(_ld(Y)), _p(Cacc), C.c_int(_ld(Cacc)),
                             C.c_int(M), C.c_int(X.shape[1]), C.c_int(Y.shape[1]), C.c_int(_dt(X))), 'a4r_gemm_tn')


def gemm_tn_bias(X, Y, Cacc, xsum, M=None):
    """Cacc += X^T Y and xsum[:P] += column sums of X (a trainable Linear's dW and db from one pass over dy)."""
    require_gpu(X, Y, Cacc, xsum)
    assert Cacc.dtype == torch.float32 and xsum.dtype == torch.float32 and xsum.numel() >= X.shape[1] and _dt(X) == _dt(Y)
    M = X.shape[0] if M is None else M
    _check(lib().a4r_gemm_tn_bias(_stream(), _p(X), C.c_int(_ld(X)), _p(Y), C.c_int(_ld(Y)), _p(Cacc), C.c_int(_ld(Cacc)),
                                  C.c_int(M), C.c_int(X.shape[1]), C.c_int(Y.shape[1]), C.c_int(_dt(X)), _p(xsum)), 'a4r_gemm_tn_bias')


class TnProb(C.Structure):
    _fields_ = [('X', C.c_void_p), ('Y', C.c_void_p), ('C', C.c_void_p), ('xsum', C.c_void_p),
                ('ldx', C.c_int32), ('ldy', C.c_int32), ('ldc', C.c_int32), ('P', C.c_int32), ('Q', C.c_int32), ('pad_', C.c_int32)]


def gemm_tn_multi(probs, M=None):
    """probs: 1..4 tuples (X, Y, Cacc, xsum or None) over the same M rows: Cacc += X^T Y, xsum += column sums of X (include/a4r.h: a4r_gemm_tn_multi)."""
    assert 1 <= len(probs) <= 4
    arr = (TnProb * len(probs))()
    dt = _dt(probs[0][0])
    M = probs[0][0].shape[0] if M is None else M
    for a, (X, Y, Cacc, xsum) in zip(arr, probs):
        require_gpu(X, Y, Cacc, xsum)
        assert Cacc.dtype == torch.float32 and _dt(X) == dt and _dt(Y) == dt and (xsum is None or (xsum.dtype == torch.float32 and xsum.numel() >= X.shape[1]))
        a.X, a.Y, a.C, a.xsum = X.data_ptr(), Y.data_ptr(), Cacc.data_ptr(), (xsum.data_ptr() if xsum is not None else None)
        a.ldx, a.ldy, a.ldc, a.P, a.Q = _ld(X), _ld(Y), _ld(Cacc), X.shape[1], Y.shape[1]
    _check(lib().a4r_gemm_tn_multi(_stream(), arr, C.c_int(len(probs)), C.c_int(M), C.c_int(dt)), 'a4r_gemm_tn_multi')


def gemm_tn2(X1, Y1, C1, X2, Y2, C2, M=None, xsum1=None, xsum2=None):
    """C1 += X1^T Y1 and C2 += X2^T Y2 over the same M rows, one launch (bf16; equal tile counts); xsum_k[p] += column sums of X_k."""
    require_gpu(X1, Y1, C1, X2, Y2, C2, xsum1, xsum2)
    assert C1.dtype == torch.float32 and C2.dtype == torch.float32
    assert (xsum1 is None or (xsum1.dtype == torch.float32 and xsum1.numel() >= X1.shape[1])) and (xsum2 is None or (xsum2.dtype == torch.float32 and xsum2.numel() >= X2.shape[1]))
    M = X1.shape[0] if M is None else M
    _check(lib().a4r_gemm_tn2(_stream(), _p(X1), C.c_int(_ld(X1)), _p(Y1), C.c_int(_ld(Y1)), _p(C1), C.c_int(_ld(C1)), C.c_int(X1.shape[1]), C.c_int(Y1.shape[1]),
                              _p(X2), C.c_int(_ld(X2)), _p(Y2), C.c_int(_ld(Y2)), _p(C2), C.c_int(_ld(C2)), C.c_int(X2.shape[1]), C.c_int(Y2.shape[1]),
                              C.c_int(M), C.c_int(_dt(X1)), _p(xsum1), _p(xsum2)), 'a4r_gemm_tn2')


def colsum(X, out, M=None):
    require_gpu(X, out)
    M = X.shape[0] if M is None else M
    _check(lib().a4r_colsum(_stream(), _p(X), C.c_int(_ld(X)), _p(out), C.c_int(M), C.c_int(X.shape[1]), C.c_int(_dt(X))), 'a4r_colsum')


def _attn_args(qkv, q_off, k_off, v_off, key_mask, n_items, S, n_heads, dh, causal, scale, mask_neg, drop_p, drop_site, drop_seed, offsets=None):
    a = AttnArgs()
    if offsets is not None:
        require_gpu(offsets)
        assert offsets.dtype == torch.int32 and offsets.numel() >= n_items + 1
    a.offsets = _p(offsets)
    a.qkv, a.ld, a.q_off, a.k_off, a.v_off = _p(qkv), _ld(qkv), q_off, k_off, v_off
    a.key_mask = _p(key_mask)
    a.n_items, a.S, a.n_heads, a.dh, a.causal, a.dtype = n_items, S, n_heads, dh, int(causal), _dt(qkv)
    a.scale, a.mask_neg = scale, mask_neg
    a.drop_p, a.drop_site, a.drop_seed = drop_p, drop_site, drop_seed
    return a


def attn_fwd(qkv, out, key_mask, n_items, S, n_heads, dh, q_off, k_off, v_off, causal, scale, mask_neg,
             drop_p=0.0, drop_site=0, drop_seed=0, offsets=None):
    """offsets (int32 [n_items + 1] on the device): packed items, see a4r_attn_t.offsets"""
    require_gpu(qkv, out)
    a = _attn_args(qkv, q_off, k_off, v_off, key_mask, n_items, S, n_heads, dh, causal, scale, mask_neg, drop_p, drop_site, drop_seed, offsets)
    a.out, a.ldo = _p(out), _ld(out)
    _check(lib().a4r_attn_fwd(_stream(), C.byref(a)), 'a4r_attn_fwd')


def attn_bwd(qkv, dout, dqkv, key_mask, n_items, S, n_heads, dh, q_off, k_off, v_off, causal, scale, mask_neg,
             drop_p=0.0, drop_site=0, drop_seed=0, offsets=None):
    require_gpu(qkv, dout, dqkv)
    assert _ld(dqkv) == _ld(qkv)
    a = _attn_args(qkv, q_off, k_off, v_off, key_mask, n_items, S, n_heads, dh, causal, scale, mask_neg, drop_p, drop_site, drop_seed, offsets)
    a.dout, a.ldo, a.dqkv = _p(dout), _ld(dout), _p(dqkv)
    _check(lib().a4r_attn_bwd(_stream(), C.byref(a)), 'a4r_attn_bwd')


def attn_long_fwd(qkv, out, lse, n_items, S, n_heads, dh, q_off, k_off, v_off, scale, drop_p=0.0, drop_site=0, drop_seed=0, key_mask=None, causal=False):
    """key_mask (fp32 [n_items, S], 1 = attend; optional, head width 64): HF's attention_mask -- text towers with more than 32 tokens per title"""
    require_gpu(qkv, out, lse, key_mask)
    assert lse.dtype == torch.float32 and lse.numel() >= n_items * n_heads * S
    a = _attn_args(qkv, q_off, k_off, v_off, key_mask, n_items, S, n_heads, dh, causal, scale, 0.0, drop_p, drop_site, drop_seed)
    a.out, a.ldo = _p(out), _ld(out)
    _check(lib().a4r_attn_long_fwd(_stream(), C.byref(a), _p(lse)), 'a4r_attn_long_fwd')


def attn_long_bwd(qkv, out, dout, dqkv, lse, delta_ws, n_items, S, n_heads, dh, q_off, k_off, v_off, scale,
                  drop_p=0.0, drop_site=0, drop_seed=0, key_mask=None, causal=False):
    """out: the ctx attn_long_fwd wrote (backward takes delta = dO . O from it)."""
    require_gpu(qkv, out, dout, dqkv, lse, delta_ws, key_mask)
    assert _ld(dqkv) == _ld(qkv) and delta_ws.dtype == torch.float32 and delta_ws.numel() >= n_items * n_heads * S
    a = _attn_args(qkv, q_off, k_off, v_off, key_mask, n_items, S, n_heads, dh, causal, scale, 0.0, drop_p, drop_site, drop_seed)
    assert _ld(out) == _ld(dout)
    a.out, a.dout, a.ldo, a.dqkv = _p(out), _p(dout), _ld(dout), _p(dqkv)
    _check(lib().a4r_attn_long_bwd(_stream(), C.byref(a), _p(lse), _p(delta_ws)), 'a4r_attn_long_bwd')


def encoder_layer_fwd(desc, x, x1, x_out):
    """a4r_encoder_layer_fwd: the 7 launches of one post-LN encoder layer with serial Houlsby adapters from ONE call (desc: EncoderLayer)."""
    require_gpu(x, x1, x_out)
    _check(lib().a4r_encoder_layer_fwd(_stream(), C.byref(desc), _p(x), _p(x1), _p(x_out)), 'a4r_encoder_layer_fwd')


def encoder_layer_bwd(desc, x1, x_out, dx_out, dx_in):
    """a4r_encoder_layer_bwd: its 9 backward launches (dx_in None: the d qkv product is skipped)."""
    require_gpu(x1, x_out, dx_out, dx_in)
    _check(lib().a4r_encoder_layer_bwd(_stream(), C.byref(desc), _p(x1), _p(x_out), _p(dx_out), _p(dx_in)), 'a4r_encoder_layer_bwd')


def patchify(img, out, patch, keep_idx=None):
    """img fp32 [n, C, H, W] (normalised) or uint8 [n, H, W, C] (raw) -> out [n * n_keep, >= C*patch*patch]."""
    require_gpu(img, out)
    assert img.is_contiguous()
    if img.dtype == torch.uint8:
        kind, (n, Hi, Wi, Cc) = 1, img.shape
    else:
        assert img.dtype == torch.float32
        kind, (n, Cc, Hi, Wi) = 0, img.shape
    n_keep = keep_idx.shape[1] if keep_idx is not None else (Hi // patch) * (Wi // patch)
    assert keep_idx is None or (keep_idx.dtype == torch.int32 and keep_idx.is_contiguous() and keep_idx.shape[0] == n)
    _check(lib().a4r_patchify(_stream(), _p(img), C.c_int(kind), _p(out), C.c_int(_ld(out)), _p(keep_idx), C.c_int(n_keep),
                              C.c_int(n), C.c_int(Cc), C.c_int(Hi), C.c_int(Wi), C.c_int(patch), C.c_int(_dt(out))), 'a4r_patchify')


def mae_keep_indices(keep, n_patches, noise=None, seed=0, site=0):
    """keep int32 [n_items, n_keep] <- argsort(noise, 1)[:, :n_keep] (stable); noise None: counter-hash uniform noise drawn on the device."""
    require_gpu(keep, noise)
    assert keep.dtype == torch.int32 and keep.is_contiguous()
    assert noise is None or (noise.dtype == torch.float32 and noise.is_contiguous() and tuple(noise.shape) == (keep.shape[0], n_patches))
    _check(lib().a4r_mae_keep_indices(_stream(), _p(noise), _p(keep), C.c_int(keep.shape[0]), C.c_int(n_patches), C.c_int(keep.shape[1]),
                                      C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(site)), 'a4r_mae_keep_indices')


def resample_u8(src, dst, bounds, kk, n_outer, in_len, out_len, inner):
    require_gpu(src, dst, bounds, kk)
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and bounds.dtype == torch.int32 and kk.dtype == torch.int32
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == n_outer * in_len * inner and dst.numel() == n_outer * out_len * inner
    _check(lib().a4r_resample_u8(_stream(), _p(src), _p(dst), _p(bounds), _p(kk), C.c_int(kk.shape[1]), C.c_long(n_outer),
                                 C.c_int(in_len), C.c_int(out_len), C.c_long(inner)), 'a4r_resample_u8')


def vit_assemble(patches, cls, pos, out, n_items, n_keep, keep_idx=None, tokens_out=0):
    require_gpu(patches, out)
    _check(lib().a4r_vit_assemble(_stream(), _p(patches), C.c_int(_ld(patches)), _p(cls), _p(pos), _p(keep_idx), _p(out),
                                  C.c_int(_ld(out)), C.c_int(n_items), C.c_int(n_keep), C.c_int(cls.numel()), C.c_int(_dt(out)),
                                  C.c_int(tokens_out)), 'a4r_vit_assemble')


def embed_bwd(ids, dpre, dword, dpos, n_items, S, roberta=False, pad_id=0):
    require_gpu(ids, dpre)
    _check(lib().a4r_embed_bwd(_stream(), _p(ids), C.c_int(ids.stride(0)), _p(dpre), C.c_int(_ld(dpre)), _p(dword), _p(dpos),
                               C.c_int(n_items), C.c_int(S), C.c_int(dpre.shape[1]), C.c_int(int(roberta)), C.c_int(pad_id),
                               C.c_int(_dt(dpre))), 'a4r_embed_bwd')


def embed_ln(ids, word, pos, type0, gamma, beta, eps, out, n_items, S, roberta=False, pad_id=0,
             drop_p=0.0, drop_site=0, drop_seed=0, pre_out=None, stats_out=None, key_mask_out=None):
    require_gpu(ids, word, out)
    assert ids.dtype == torch.int64 and ids.stride(1) == 1
    _check(lib().a4r_embed_ln(_stream(), _p(ids), C.c_int(ids.stride(0)), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta),
                              C.c_float(eps), _p(out), C.c_int(_ld(out)), C.c_int(n_items), C.c_int(S), C.c_int(word.shape[1]),
                              C.c_int(int(roberta)), C.c_int(pad_id), C.c_int(_dt(out)),
                              C.c_float(drop_p), C.c_uint32(drop_site), C.c_uint64(drop_seed), _p(pre_out), _p(stats_out), _p(key_mask_out)), 'a4r_embed_ln')


def quantize_weight_fp8(w):
    """Frozen weight [out, in] -> (e4m3 bit patterns uint8 [out, in], fp32 scale [out]): per-output-channel absmax scaling, the B
    operand form of the fp8 a4r_gemm_nt.  Build-time only (once per frozen matrix)."""
    wf = w.detach().float()
    amax = wf.abs().amax(1).clamp_min(1e-30)
    q = (wf * (448.0 / amax)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return q, (amax / 448.0).contiguous()


def quant_rows_fp8(x, q, scale, M=None):
    require_gpu(x, q, scale)
    M = x.shape[0] if M is None else M
    assert q.dtype == torch.uint8 and scale.dtype == torch.float32
    _check(lib().a4r_quant_rows_fp8(_stream(), _p(x), C.c_int(_ld(x)), _p(q), C.c_int(_ld(q)), _p(scale), C.c_int(M), C.c_int(x.shape[1]),
                                    C.c_int(_dt(x))), 'a4r_quant_rows_fp8')


def ln_fwd_sum(h, res, gamma, beta, eps, y, stats, M=None, res32=None, sum_out=None, sum32=None, y32=None):
    """y = LN(h + residual), sum in fp32, normalised unrounded; residual = res32 (fp32) when given, else res (h's dtype).  Optional outputs:
    sum_out (h's dtype), sum32, y32 (fp32)."""
    require_gpu(h, res, res32, y, sum_out, sum32, y32)
    M = h.shape[0] if M is None else M
    assert res32 is not None or res is not None
    _check(lib().a4r_ln_fwd_sum(_stream(), _p(h), C.c_int(_ld(h)), _p(res32), C.c_int(_ld(res32) if res32 is not None else 0), _p(res),
                                C.c_int(_ld(res) if res is not None else 0), _p(gamma), _p(beta), C.c_float(eps), _p(y), C.c_int(_ld(y)),
                                _p(sum_out), C.c_int(_ld(sum_out) if sum_out is not None else 0), _p(sum32), C.c_int(_ld(sum32) if sum32 is not None else 0),
                                _p(y32), C.c_int(_ld(y32) if y32 is not None else 0), _p(stats), C.c_int(M), C.c_int(h.shape[1]), C.c_int(_dt(h))),
           'a4r_ln_fwd_sum')


def ln_fwd(v, gamma, beta, eps, y, stats, M=None, add=None, drop_p=0.0, drop_site=0, drop_seed=0, y8=None, ys=None):
    require_gpu(v, y, y8)
    M = v.shape[0] if M is None else M
    if y8 is not None:                    # also emit the row as e4m3 + per-row scale (y may be None)
        assert drop_p == 0.0 and y8.dtype == torch.uint8 and ys.dtype == torch.float32
        _check(lib().a4r_ln_fwd_fp8(_stream(), _p(v), C.c_int(_ld(v)), _p(add), C.c_int(add.shape[0] if add is not None else 0),
                                    _p(gamma), _p(beta), C.c_float(eps), _p(y), C.c_int(_ld(y) if y is not None else 0), _p(y8), C.c_int(_ld(y8)),
                                    _p(ys), _p(stats), C.c_int(M), C.c_int(v.shape[1]), C.c_int(_dt(v))), 'a4r_ln_fwd_fp8')
        return
    _check(lib().a4r_ln_fwd(_stream(), _p(v), C.c_int(_ld(v)), _p(add), C.c_int(add.shape[0] if add is not None else 0),
                            _p(gamma), _p(beta), C.c_float(eps), _p(y), C.c_int(_ld(y)), _p(stats), C.c_int(M),
                            C.c_int(v.shape[1]), C.c_int(_dt(v)), C.c_float(drop_p), C.c_uint32(drop_site), C.c_uint64(drop_seed)),
           'a4r_ln_fwd')


def ln_bwd(dy, v, stats, gamma, dv, M=None, add=None, dgamma=None, dbeta=None, dbias=None, dres=None,
           drop_p=0.0, drop_site=0, drop_seed=0, dv2=None, drop2_p=0.0, drop2_site=0, drop2_seed=0):
    require_gpu(dy, v, dv)
    M = v.shape[0] if M is None else M
    _check(lib().a4r_ln_bwd(_stream(), _p(dy), C.c_int(_ld(dy)), _p(v), C.c_int(_ld(v)), _p(add),
                            C.c_int(add.shape[0] if add is not None else 0), _p(stats), _p(gamma), _p(dres), C.c_int(_ld(dres) if dres is not None else 0), _p(dv), C.c_int(_ld(dv)),
                            _p(dgamma), _p(dbeta), _p(dbias), C.c_int(M), C.c_int(v.shape[1]), C.c_int(_dt(v)),
                            C.c_float(drop_p), C.c_uint32(drop_site), C.c_uint64(drop_seed),
                            _p(dv2), C.c_int(_ld(dv2) if dv2 is not None else 0), C.c_float(drop2_p), C.c_uint32(drop2_site), C.c_uint64(drop2_seed)), 'a4r_ln_bwd')


def gather_rows(src, dst, n, row_step):
    require_gpu(src, dst)
    _check(lib().a4r_gather_rows(_stream(), _p(src), C.c_int(_ld(src)), _p(dst), C.c_int(_ld(dst)), C.c_int(n), C.c_int(row_step),
                                 C.c_int(src.shape[1]), C.c_int(_dt(src))), 'a4r_gather_rows')


def rows_idx_copy(src, dst, idx, n, scatter=False):
    """dst[r] = src[idx[r]] (scatter: dst[idx[r]] = src[r]) for r < n; 2-D tensors of one dtype, row bytes a multiple of 16; idx int32 on the device."""
    require_gpu(src, dst, idx)
    assert src.dim() == 2 and dst.dim() == 2 and src.dtype == dst.dtype and src.shape[1] == dst.shape[1] and idx.dtype == torch.int32 and idx.numel() >= n
    es = src.element_size()
    _check(lib().a4r_rows_idx_copy(_stream(), _p(src), C.c_int64(src.stride(0) * es), _p(dst), C.c_int64(dst.stride(0) * es), _p(idx), C.c_int(n),
                                   C.c_int64(src.shape[1] * es), C.c_int(int(scatter))), 'a4r_rows_idx_copy')


def scatter_rows(src, dst, n, row_step):
    require_gpu(src, dst)
    _check(lib().a4r_scatter_rows(_stream(), _p(src), C.c_int(_ld(src)), _p(dst), C.c_int(_ld(dst)), C.c_int(n), C.c_int(row_step),
                                  C.c_int(src.shape[1]), C.c_int(_dt(src))), 'a4r_scatter_rows')


def scatter_rows_fill(src, dst, n, row_step, fill_rows):
    """dst[r] = src[r / row_step] for r % row_step == 0 (r / row_step < n), 0 elsewhere, r < fill_rows."""
    require_gpu(src, dst)
    _check(lib().a4r_scatter_rows_fill(_stream(), _p(src), C.c_int(_ld(src)), _p(dst), C.c_int(_ld(dst)), C.c_int(n), C.c_int(row_step),
                                       C.c_int(src.shape[1]), C.c_int(_dt(src)), C.c_int(fill_rows)), 'a4r_scatter_rows_fill')


def zero(t):
    """hipMemsetAsync of a contiguous tensor on the current stream."""
    require_gpu(t)
    assert t.is_contiguous()
    _check(lib().a4r_memset_zero(_stream(), _p(t), C.c_int64(t.numel() * t.element_size())), 'a4r_memset_zero')


def lora_merge(W, A, B, scaling, dst, dstT, r):
    """dst [out, in] (a row block of the packed qkv operand) and dstT [in, out] (a column block of its transpose) <- W + scaling B A."""
    require_gpu(W, dst, dstT)
    out_f, in_f = W.shape
    assert W.dtype == torch.float32 and W.is_contiguous() and (r == 0 or (A.is_contiguous() and B.is_contiguous()))
    _check(lib().a4r_lora_merge(_stream(), _p(W), _p(A) if r else C.c_void_p(0), _p(B) if r else C.c_void_p(0), C.c_float(scaling),
                                _p(dst), C.c_int(_ld(dst)), _p(dstT), C.c_int(_ld(dstT)), C.c_int(out_f), C.c_int(in_f), C.c_int(r),
                                C.c_int(_dt(dst))), 'a4r_lora_merge')


class LoraDesc(C.Structure):
    _fields_ = [('W', C.c_void_p), ('A', C.c_void_p), ('B', C.c_void_p), ('dst', C.c_void_p), ('dstT', C.c_void_p),
                ('scaling', C.c_float), ('ld', C.c_int32), ('ldT', C.c_int32), ('out_f', C.c_int32), ('in_f', C.c_int32), ('r', C.c_int32)]


def lora_table(entries, device):
    """entries: (W, A, B, scaling, dst, dstT, r) per projection (the arguments of lora_merge) -> (device table, n, max elements, dtype code)."""
    descs = []
    for W, A, B, s, dst, dstT, r in entries:
        assert W.dtype == torch.float32 and W.is_contiguous() and (r == 0 or (A.is_contiguous() and B.is_contiguous())) and _dt(dst) == _dt(entries[0][4])
        descs.append(LoraDesc(W.data_ptr(), A.data_ptr() if r else 0, B.data_ptr() if r else 0, dst.data_ptr(), dstT.data_ptr(), float(s),
                              _ld(dst), _ld(dstT), W.shape[0], W.shape[1], int(r)))
    return desc_table(descs, device), len(descs), max(W.numel() for W, *_ in entries), _dt(entries[0][4])


def lora_merge_batch(tab):
    t, n, mx, code = tab
    _check(lib().a4r_lora_merge_batch(_stream(), _p(t), C.c_int(n), C.c_int(mx), C.c_int(code)), 'a4r_lora_merge_batch')


def lora_bwd_fused_ok(x, M, H):
    """the geometry a4r_lora_bwd_fused is built for (everything else keeps the separate products)"""
    return x.dtype == torch.bfloat16 and H == 768 and M % 16 == 0


_lora_ws = {}


def lora_bwd_fused(x, dqa, dqb, Aa, Ab, BTa, BTb, scale_a, scale_b, dAa, dAb, dBa, dBb, dbias_a, dbias_b, M, rank_rows=8):
    """One pass over x, dqa, dqb: dAa | dAb += dt^T x, dBa += dqa^T t_a, dBb += dqb^T t_b (unscaled), dbias_. += column sums (include/a4r.h).  The
    weight operands are views of rank_rows (8, or 16 for ranks 9 - 15) rank rows, the outputs views into the fp32 scratch matrices the corners are
    flushed from."""
    require_gpu(x, dqa, dqb, dAa, dBa)
    H = x.shape[1]
    assert _ld(dqa) == _ld(dqb) and _ld(Aa) == _ld(Ab) == _ld(BTa) == _ld(BTb) and _ld(dAa) == _ld(dAb) and _ld(dBa) == _ld(dBb)
    ldbias = 0
    for bvec in (dbias_a, dbias_b):
        if bvec is not None:
            assert bvec.dtype == torch.float32 and bvec.dim() == 1
            ldbias = bvec.stride(0)
    ws = _lora_ws.get(x.device)                      # the workgroups' column sums before their reduction: one buffer per device (stream-ordered reuse)
    if ws is None:
        ws = _lora_ws[x.device] = torch.empty(int(lib().a4r_lora_bwd_fused_ws_floats(C.c_int(H))), dtype=torch.float32, device=x.device)
    _check(lib().a4r_lora_bwd_fused(_stream(), _p(x), C.c_int(_ld(x)), _p(dqa), _p(dqb), C.c_int(_ld(dqa)), _p(Aa), _p(Ab), _p(BTa), _p(BTb),
                                    C.c_int(_ld(Aa)), C.c_float(scale_a), C.c_float(scale_b), _p(dAa), _p(dAb), C.c_int(_ld(dAa)), _p(dBa), _p(dBb),
                                    C.c_int(_ld(dBa)), _p(dbias_a), _p(dbias_b), C.c_int(ldbias), C.c_int(M), C.c_int(H), C.c_int(_dt(x)),
                                    C.c_int(rank_rows), _p(ws), C.c_int64(ws.numel())), 'a4r_lora_bwd_fused')


def desc_table(entries, device):
    """ctypes descriptor array -> device byte tensor."""
    arr = (type(entries[0]) * len(entries))(*entries)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def phm_build(params, desc_dev, n_desc, eff):
    _check(lib().a4r_phm_build(_stream(), _p(params), _p(desc_dev), C.c_int(n_desc), _p(eff)), 'a4r_phm_build')


def phm_bwd(params, desc_dev, n_desc, grads):
    _check(lib().a4r_phm_bwd(_stream(), _p(params), _p(desc_dev), C.c_int(n_desc), _p(grads)), 'a4r_phm_bwd')


def unpack_add(target, desc_dev, n_desc, max_elems):
    _check(lib().a4r_unpack_add(_stream(), _p(target), _p(desc_dev), C.c_int(n_desc), C.c_int(max_elems)), 'a4r_unpack_add')


def dropout_apply(x, y, drop_p, drop_site, drop_seed, M=None):
    require_gpu(x, y)
    M = x.shape[0] if M is None else M
    _check(lib().a4r_dropout_apply(_stream(), _p(x), C.c_int(_ld(x)), _p(y), C.c_int(_ld(y)), C.c_int(M), C.c_int(x.shape[1]),
                                   C.c_int(_dt(x)), C.c_float(drop_p), C.c_uint32(drop_site), C.c_uint64(drop_seed)), 'a4r_dropout_apply')


def act_bwd_f32(dy, pre, dx, act):
    require_gpu(dy, pre, dx)
    _check(lib().a4r_act_bwd_f32(_stream(), _p(dy), _p(pre), _p(dx), C.c_int64(dy.numel()), C.c_int(act)), 'a4r_act_bwd_f32')


def score_bce_fwd(emb, prec, log_mask, pos, neg, loss_ws, B, L, E, cpc):
    require_gpu(emb, prec)
    _check(lib().a4r_score_bce_fwd(_stream(), _p(emb), _p(prec), _p(log_mask), _p(pos), _p(neg), _p(loss_ws),
                                   C.c_int(B), C.c_int(L), C.c_int(E), C.c_int(int(cpc))), 'a4r_score_bce_fwd')


def score_bce_bwd(emb, prec, log_mask, pos, neg, loss_ws, loss_scale, d_prec, d_emb, B, L, E, cpc, scale_dev=None):
    require_gpu(emb, prec, scale_dev)
    assert scale_dev is None or (scale_dev.dtype == torch.float32 and scale_dev.numel() == 1)
    _check(lib().a4r_score_bce_bwd(_stream(), _p(emb), _p(prec), _p(log_mask), _p(pos), _p(neg), _p(loss_ws), C.c_float(loss_scale), _p(scale_dev),
                                   _p(d_prec), _p(d_emb), C.c_int(B), C.c_int(L), C.c_int(E), C.c_int(int(cpc))), 'a4r_score_bce_bwd')


def emb_grad_add_inputs(d_in, d_emb, B, L, E):
    _check(lib().a4r_emb_grad_add_inputs(_stream(), _p(d_in), C.c_int(_ld(d_in)), _p(d_emb), C.c_int(B), C.c_int(L), C.c_int(E)),
           'a4r_emb_grad_add_inputs')


def take_inputs(emb, out, B, L, E):
    _check(lib().a4r_take_inputs(_stream(), _p(emb), _p(out), C.c_int(_ld(out)), C.c_int(B), C.c_int(L), C.c_int(E)), 'a4r_take_inputs')


def adam_step(p, g, m, v, seg_end, seg_group, group_lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    require_gpu(p, g, m, v, seg_end, seg_group, group_lr)
    _check(lib().a4r_adam_step(_stream(), _p(p), _p(g), _p(m), _p(v), C.c_int64(p.numel()), _p(seg_end), _p(seg_group),
                               C.c_int(seg_end.numel()), _p(group_lr), C.c_int(step), C.c_float(beta1), C.c_float(beta2),
                               C.c_float(eps), C.c_float(grad_scale)), 'a4r_adam_step')


def pack_matrices(flat, desc_dev, n_desc, max_elems, dtype):
    _check(lib().a4r_pack_matrices(_stream(), _p(flat), _p(desc_dev), C.c_int(n_desc), C.c_int(max_elems), C.c_int(dtype)),
           'a4r_pack_matrices')


def eval_rank(prec, item_emb, target, hist_ptr, hist_idx, rank):
    require_gpu(prec, item_emb, target, hist_ptr, hist_idx, rank)
    _check(lib().a4r_eval_rank(_stream(), _p(prec), _p(item_emb), _p(target), _p(hist_ptr), _p(hist_idx), _p(rank),
                               C.c_int(prec.shape[0]), C.c_int(item_emb.shape[0]), C.c_int(prec.shape[1])), 'a4r_eval_rank')
