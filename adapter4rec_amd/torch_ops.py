"""The TORCH_LIBRARY form of the boundary: ``torch.ops.a4r.*`` -- at::Tensor arguments, TORCH_CHECK errors, kernels
enqueued on the current HIP stream -- registered by ``liba4r_torch_ops.so`` (adapter4rec_amd/csrc/a4r_torch_ops.cpp), a host-only shim over
the C ABI of ``liba4r_hip.so``.  The training path itself binds the C ABI through ctypes (``_lib.py``); this module is for C++ /
TorchScript / dispatcher-level callers and for the tests that hold the two bindings to each other.

    from adapter4rec_amd import torch_ops
    ops = torch_ops.load()                                   # raises when the library has not been built
    ops.gemm_nt(x, w, y, bias, None, None, 0, 1.0, 0.0, 0, 0, False)
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_LIB_PATH = os.path.join(_HERE, 'liba4r_torch_ops.so')
OPS = ('gemm_nt', 'adapter_residual_ln_fwd', 'adapter_residual_ln_bwd', 'ln_fwd', 'score_bce_fwd', 'score_bce_bwd', 'fused_adam_step',
       'topk_rank_eval', 'lora_bwd', 'encoder_layer_fwd', 'encoder_layer_bwd', 'sasrec_block_fwd', 'sasrec_block_bwd', 'embed_ln_fwd', 'patch_embed_fwd',
       'vit_assemble', 'abi_version')
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f'{OPS_LIB_PATH} not found: build it with `make -C adapter4rec_amd/csrc` (python -c "import __graft_entry__ as g; g.build()")')
        torch.ops.load_library(OPS_LIB_PATH)
        from . import _lib
        got = int(torch.ops.a4r.abi_version())
        if got != _lib.ABI_VERSION:
            raise RuntimeError(f'{OPS_LIB_PATH} was built against ABI {got}, this package is {_lib.ABI_VERSION}: rebuild')
        _loaded = True
    return torch.ops.a4r


# ---------------------------------------------------------------------------------------------------------------- autograd formulas
# (round 6) The ops above are out-variants with caller-owned workspaces, as SURVEY 8(b) asks ("workspace passed in by caller"); the two classes below are
# their autograd forms for a caller that wants a differentiable module instead: they own the saved-for-backward tensors, return what the C entry points
# compute, and hand autograd the gradients of the layer input and of the TRAINABLE adapter parameters (the backbone is frozen: dgrad only, None for it).

class EncoderLayerFunction(torch.autograd.Function):
    """One post-LN encoder layer with serial Houlsby adapters on both halves (HF BertLayer + BertAdaptedSelfOutput, Downstream/Text/model/model.py:292-297)
    over torch.ops.a4r.encoder_layer_fwd / _bwd.

        y = EncoderLayerFunction.apply(x, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2, frozen, key_mask, cfg)

    x bf16 [M, H] (M % 128 == 0, rows >= n_items * S zero); adapter parameters fp32: wd [64, H], bd [64], wu [H, 64], bu [H]; frozen = the 12 tensors
    (wqkv, bqkv, wo, bo, wi, bi, wo2, bo2, ln1_g, ln1_b, ln2_g, ln2_b: bf16 [out, in] weights, fp32 vectors); key_mask fp32 [n_items, S] or None;
    cfg = dict(n_items, S, n_heads, act1, act2, ln_eps=1e-12, p_attn=0, p_hidden=0, drop_site=0, drop_seed=0, mask_neg=finfo(float32).min).
    Returns y bf16 [M, H]; backward: d x (bf16) and fp32 gradients of the eight adapter tensors."""

    @staticmethod
    def forward(ctx, x, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2, frozen, key_mask, cfg):
        import math
        ops = load()
        M, H = x.shape
        F = frozen[4].shape[0]
        t, dev = torch.bfloat16, x.device
        mk = lambda c, dt=t: torch.zeros(M, c, dtype=dt, device=dev)
        ad = [[wd1.detach().to(t).contiguous(), bd1.detach().float().contiguous(), wu1.detach().to(t).contiguous(), bu1.detach().float().contiguous()],
              [wd2.detach().to(t).contiguous(), bd2.detach().float().contiguous(), wu2.detach().to(t).contiguous(), bu2.detach().float().contiguous()]]
        saved = [mk(3 * H), mk(H), mk(H), mk(H), mk(64), mk(64), mk(F), mk(F, torch.uint8), mk(H), mk(H), mk(64), mk(64), mk(2, torch.float32), mk(2, torch.float32)]
        x1, y = mk(H), mk(H)
        sc = dict(n_items=int(cfg['n_items']), S=int(cfg['S']), n_heads=int(cfg['n_heads']), causal=False, scale=1.0 / math.sqrt(H // int(cfg['n_heads'])),
                  mask_neg=float(cfg.get('mask_neg', torch.finfo(torch.float32).min)), ln_eps=float(cfg.get('ln_eps', 1e-12)), p_attn=float(cfg.get('p_attn', 0.0)),
                  p_hidden=float(cfg.get('p_hidden', 0.0)), drop_site=int(cfg.get('drop_site', 0)), drop_seed=int(cfg.get('drop_seed', 0)),
                  act1=int(cfg['act1']), act2=int(cfg['act2']))
        ops.encoder_layer_fwd(x, list(frozen), ad[0], ad[1], saved, x1, y, key_mask, None, q8_tiled=False, **sc)
        ctx.sc, ctx.frozen, ctx.key_mask, ctx.ad, ctx.saved, ctx.x1 = sc, list(frozen), key_mask, ad, saved, x1
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = load()
        (y,) = ctx.saved_tensors
        M, H = y.shape
        F = ctx.frozen[4].shape[0]
        t, dev = torch.bfloat16, y.device
        mk = lambda c, dt=t: torch.zeros(M, c, dtype=dt, device=dev)
        w = ctx.frozen
        wT = [w[0].t().contiguous(), w[2].t().contiguous(), w[4].t().contiguous(), w[6].t().contiguous()]
        adT = [[a[0].t().contiguous(), a[2].t().contiguous()] for a in ctx.ad]
        scratch = [mk(H), mk(H), mk(64), mk(H), mk(F), mk(H), mk(H), mk(3 * H)]
        grads = [[torch.zeros(H, 64, device=dev), torch.zeros(64, H, device=dev), torch.zeros(H, device=dev), torch.zeros(64, device=dev)] for _ in range(2)]
        dx = mk(H)
        dyc = dy.to(t).contiguous().clone()
        dyc[ctx.sc['n_items'] * ctx.sc['S']:] = 0                  # (padding rows carry no gradient)
        ops.encoder_layer_bwd(dyc, ctx.x1, y, w, wT, ctx.ad[0], ctx.ad[1], adT[0], adT[1], ctx.saved, scratch, grads[0], grads[1], dx, ctx.key_mask, None,
                              q8_tiled=False, **ctx.sc)
        (gu1, gd1, gbu1, gbd1), (gu2, gd2, gbu2, gbd2) = grads
        return dx, gd1, gbd1, gu1, gbu1, gd2, gbd2, gu2, gbu2, None, None, None


class SasrecBlockFunction(torch.autograd.Function):
    """One SASRec block with its two adapters (TransformerBlock + SASRecAdaptedSelfOutput, Downstream/Text/model/modules.py:45-87, model.py:341-376) over
    torch.ops.a4r.sasrec_block_fwd / _bwd.

        y = SasrecBlockFunction.apply(x, log_mask, frozen10, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2, cfg)

    x fp32 [B, T, 64]; log_mask fp32 [B, T]; frozen10 = (wqkv, wfc, w1, b1, w2, b2, ln1_g, ln1_b, ln2_g, ln2_b), fp32; adapters fp32: wd [dp, 64], bd [dp],
    wu [64, dp], bu [64] (dp = the bottleneck rounded up to 16, zero-padded); cfg = dict(n_heads, F, d, act, inner_res, eps, mask_neg)."""

    @staticmethod
    def forward(ctx, x, log_mask, frozen, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2, cfg):
        ops = load()
        wl = [q.detach().contiguous() for q in (*frozen, wd1, bd1, wu1, bu1, wd2, bd2, wu2, bu2)]
        y = torch.zeros_like(x)
        common = dict(n_heads=int(cfg['n_heads']), F=int(cfg['F']), d=int(cfg['d']), act=int(cfg['act']), inner_res=bool(cfg['inner_res']), eps=float(cfg['eps']),
                      mask_neg=float(cfg['mask_neg']))
        ops.sasrec_block_fwd(x.contiguous(), log_mask.contiguous(), y, wl, **common)
        ctx.wl, ctx.common = wl, common
        ctx.save_for_backward(x, log_mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = load()
        x, log_mask = ctx.saved_tensors
        gl = [torch.zeros_like(q) for q in ctx.wl[10:]]
        dx = torch.zeros_like(x)
        ops.sasrec_block_bwd(x.contiguous(), log_mask.contiguous(), dy.contiguous(), dx, ctx.wl, gl, **ctx.common)
        return (dx, None, None, *gl, None)
