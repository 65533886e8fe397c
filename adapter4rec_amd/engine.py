"""MI355X-native execution engine for the adapter-tuned TransRec step.

Replaces, behind ``Model.forward`` / ``Bert_Encoder`` / ``User_Encoder`` (reference:
Downstream/Text/model/model.py:48-70, encoders.py:24-29,89-99), the eager op sequence
BERT/RoBERTa item encoder (+ Houlsby / Pfeiffer / Compacter adapters) -> SASRec/CPC user encoder
-> dot-product BCE head -> backward (dgrad-only through the frozen backbone) with launches of the
C-ABI kernels in liba4r_hip.so.  Python here only owns buffers, pointers and launch order.

Data layout in HBM
  * activations: row-major [M_pad, width], M = items * S tokens (S = title length) padded to 128 rows;
    bf16 (or fp32 in --compute_dtype fp32) for the item encoder, fp32 for the tiny SASRec side;
  * frozen weights: packed once in compute dtype, both W [out, in] (forward, "NT" GEMM operand) and
    W^T [in, out] (dgrad operand) -- 2 x 170 MB bf16 for BERT-base, never touched again;
  * trainable tensors: ONE flat fp32 buffer (parameters are views into it), one flat fp32 gradient
    buffer (what RCCL all-reduces), Adam moments alongside; kernel-side bf16/transposed/padded copies
    of the adapter matrices are refreshed by one pack launch per step.
"""
import math

import torch

from . import _lib as L
from .model.bert import backbone_geometry
from .model.modules import AdapterBlock, AdapterPfeifferBlock, HyperComplexAdapterBlock, PHMLinear

FMIN = float(torch.finfo(torch.float32).min)


def pad_to(n, m):
    return (n + m - 1) // m * m


import os as _os
TN2 = bool(int(_os.environ.get('A4R_TN2', '1')))                       # an adapter's two weight gradients in one launch (0: two a4r_gemm_tn, A/B)
TN2_BIAS = bool(int(_os.environ.get('A4R_TN2_BIAS', '1')))             # an adapter's two bias gradients from its weight-gradient launch (0: from the fused backward's end-of-launch flush, A/B)
FUSE_BD = bool(int(_os.environ.get('A4R_FUSE_BD', '1')))               # db_down from the fused adapter backward kernel (0: a4r_colsum launches, A/B)
WGRAD_STREAM = bool(int(_os.environ.get('A4R_WGRAD_STREAM', '0')))     # 1 = adapter weight gradients on a side stream (see _adapter_wgrads): +1.3 % in round 1, neutral since the GEMM's
                                                                        # late-starting workgroups use the same idle CUs (same-box 19.07 vs 18.98 ms): off by default

class _LN:
    """LayerNorm parameters (fp32) + optional gradient sinks."""

    def __init__(self, mod, eng):
        self.gamma, self.beta, self.eps = mod.weight, mod.bias, mod.eps
        self.g_gamma = eng.grad_view(mod.weight)
        self.g_beta = eng.grad_view(mod.bias)


class _Adapter:
    """One bottleneck adapter: kernel-side copies (compute dtype, bottleneck padded to 64) + gradient sinks."""

    def __init__(self, mod, width, eng, dt):
        self.kind = mod.kind
        self.act = L.ACT_BY_NAME[mod.activation_name]
        dev = eng.dev
        if isinstance(mod, HyperComplexAdapterBlock):
            self.d = mod.down_sampler.out_features
            self.virtual = (mod.down_sampler, mod.up_sampler)
            b_down, b_up = mod.down_sampler.b, mod.up_sampler.b
            for lin in self.virtual:     # gradients of these arrive through _virtual_backward
                for q in (lin.W_left, lin.W_right, lin.phm_rule):
                    if q is not None:
                        eng.grad_view(q)
        else:
            self.d = mod.fc_down.out_features
            self.virtual = None
            self.p_wd, self.p_wu = mod.fc_down.weight, mod.fc_up.weight
            b_down, b_up = mod.fc_down.bias, mod.fc_up.bias
        self.dp = pad_to(self.d, 64)
        self.width = width
        self.wd = torch.zeros(self.dp, width, dtype=dt, device=dev)      # fc_down.weight  [d, H]
        self.wdT = torch.zeros(width, self.dp, dtype=dt, device=dev)
        self.wu = torch.zeros(width, self.dp, dtype=dt, device=dev)      # fc_up.weight    [H, d]
        self.wuT = torch.zeros(self.dp, width, dtype=dt, device=dev)
        self.p_bd, self.p_bu = b_down, b_up
        self.bd = torch.zeros(self.dp, dtype=torch.float32, device=dev)  # padded copy of fc_down.bias
        self.bu = b_up                                                   # [H] fp32, used in place
        self.g_bu = eng.grad_view(b_up)
        self.g_bd = eng.grad_view(b_down)
        self.g_wd = self.g_wu = None
        # the same four matrices in the FRAGMENT order of the one-launch adapter kernels (a4r_pack_matrices layouts 1 / 2, include/a4r.h: every wave
        # instruction of those kernels' prologues then reads 1 KiB contiguous; the first tile of a launch starts ~3 us earlier).  Row-major copies stay:
        # the three-launch forms, the weight-gradient kernels' shapes and the tests read them.  (fwd: wd, wu; bwd: wuT, wdT)
        self.frag_f = self.frag_b = None
        if dt == torch.bfloat16 and self.dp == 64 and width in (128, 256, 512, 768, 1024) and _os.environ.get('A4R_ADAPTER_FRAG', '1') != '0':
            mk = lambda: torch.zeros(64 * width, dtype=dt, device=dev)
            self.frag_f, self.frag_b = (mk(), mk()), (mk(), mk())
        if self.virtual is None:
            self.g_wd, self.g_wu = eng.grad_view(self.p_wd), eng.grad_view(self.p_wu)
            eng.add_pack(self.p_wd, self.wd, False)
            eng.add_pack(self.p_wd, self.wdT, True)
            eng.add_pack(self.p_wu, self.wu, False)
            eng.add_pack(self.p_wu, self.wuT, True)
            if self.frag_f is not None:
                for p_, dst, code in self.frag_entries():
                    eng.add_pack(p_, dst, code)
        else:
            eng.add_virtual(self)
        eng.add_pack_bias(b_down, self.bd)
        # scratch for padded weight gradients (used when d < dp, or for virtual matrices)
        direct = self.virtual is None and self.d == self.dp
        self.s_wd = None if direct else eng.scratch(self.dp, width)
        self.s_wu = None if direct else eng.scratch(width, self.dp)
        self.s_bd = None if self.d == self.dp else eng.scratch(1, self.dp)[0]
        if self.virtual is None and not direct:          # the valid corners go into the flat gradient at the end of backward
            eng.add_corner(self.s_wu, self.p_wu, width, self.d)
            eng.add_corner(self.s_wd, self.p_wd, self.d, width)
        if self.s_bd is not None:
            eng.add_corner(self.s_bd.view(1, -1), b_down, 1, self.d)

    def frag_entries(self, wd_src=None, wu_src=None):
        """(source, destination, a4r_pack_desc_t.transpose code) of the four fragment-ordered copies: bit 0 transpose, bits 1-2 layout (1: [64, H], 2: [H, 64]);
        destinations are viewed with their LOGICAL shape so that the descriptor carries rows_pad / cols_pad"""
        wd_src = self.p_wd if wd_src is None else wd_src
        wu_src = self.p_wu if wu_src is None else wu_src
        W = self.width
        return [(wd_src, self.frag_f[0].view(64, W), 2), (wu_src, self.frag_f[1].view(W, 64), 4),
                (wu_src, self.frag_b[0].view(64, W), 2 | 1), (wd_src, self.frag_b[1].view(W, 64), 4 | 1)]


class _Lora:
    """LoRA on one projection (q or v) of a fused qkv weight: W_eff = W + B A / r is re-merged into the packed qkv operand
    every step (so forward and dgrad cost nothing extra); the low-rank gradients come from four skinny products in backward:
    t = x A^T, dt = (dq B) s, dB = dq^T t s, dA = dt^T x."""

    SHARE = _os.environ.get('A4R_LORA_SHARE', '1') != '0'
    ONES_COL = 32            # (shared form: rank columns 0 - 7 and 16 - 23 are in use)

    def __init__(self, mod, width, eng, dt, slot, share=None):
        """share = (dict, off): this LoRA is one of a block's two small-rank ones (r <= 16: the image tower's q, v at r = 8).  They then use
        ONE [64, width] A operand (rank rows at off .. off + r), ONE t = x A^T and ONE dA = dt^T x launch; only the products that read
        this projection's own gradient (dt = dq B, dB = dq^T t, the bias sum) stay per LoRA.  The rank was padded to 64 columns anyway."""
        self.mod, self.slot, self.width = mod, slot, width
        self.r, self.rp, self.scaling = mod.r, pad_to(mod.r, 64), float(mod.scaling)
        dev = eng.dev
        self.g_bias = eng.grad_view(mod.bias) if mod.bias is not None else None
        self.g_W = None
        self.share = None
        if self.r == 0:                  # loralib: r = 0 leaves a plain Linear whose weight stays trainable (CV run_adapter.py:394)
            self.g_W = eng.grad_view(mod.weight)
            return
        self.g_A, self.g_B = eng.grad_view(mod.lora_A), eng.grad_view(mod.lora_B)
        self.BT = torch.zeros(self.rp, width, dtype=dt, device=dev)       # lora_B^T [r, out]     (NT operand of dt = dq B)
        self.s_B = eng.scratch(width, self.rp)
        if share is not None:
            sh, off = share
            self.share, self.off = sh, off
            if 'A' not in sh:
                sh['A'] = torch.zeros(self.rp, width, dtype=dt, device=dev)
                sh['s_A'] = eng.scratch(self.rp, width)
            self.A, self.s_A = sh['A'], sh['s_A']
            eng.add_pack(mod.lora_A, sh['A'][off:off + 16], False)                   # rows off .. off + r (zero-padded to 16)
            eng.add_pack(mod.lora_B, self.BT[off:off + 16], True)                    # B^T at the same rank rows: dt lands in columns off .. off + r
            eng.add_corner(self.s_B[:, off:off + 16], mod.lora_B, width, self.r, alpha=self.scaling)
            eng.add_corner(sh['s_A'][off:off + 16], mod.lora_A, self.r, width)
            # the bias gradient (column sums of this projection's gradient) rides in the dB product: column ONES_COL of t is a column of
            # ones (a bias of the t = x A^T launch: row ONES_COL of A is zero), so (dq^T t)[:, ONES_COL] = sum over rows of dq
            if 'ones' not in sh:
                sh['ones'] = torch.zeros(self.rp, dtype=torch.float32, device=dev)
                sh['ones'][self.ONES_COL] = 1.0
            if self.g_bias is not None:
                eng.add_corner(self.s_B[:, self.ONES_COL:self.ONES_COL + 1], mod.bias, width, 1)
            return
        self.A = torch.zeros(self.rp, width, dtype=dt, device=dev)        # lora_A [r, in]        (NT operand of t = x A^T)
        eng.add_pack(mod.lora_A, self.A, False)
        eng.add_pack(mod.lora_B, self.BT, True)
        self.s_A = eng.scratch(self.rp, width)
        eng.add_corner(self.s_B, mod.lora_B, width, self.r, alpha=self.scaling)      # dB = (dq^T t) s
        eng.add_corner(self.s_A, mod.lora_A, self.r, width)

    @classmethod
    def for_block(cls, lins, width, eng, dt):
        """The _Lora objects of a block's (query, key, value) projections; two small-rank ones share their A-side launches."""
        idx = [i for i, lin in enumerate(lins) if type(lin).__name__ == 'LoRALinear']
        small = [i for i in idx if 0 < lins[i].r <= 16]
        share = {} if (cls.SHARE and len(small) == 2 and len(idx) == 2) else None
        out = []
        for i in idx:
            out.append(cls(lins[i], width, eng, dt, i, share=(share, 16 * small.index(i)) if share is not None else None))
        return out


class _Block:
    """One post-LN transformer block (BERT layer or SASRec block) with optional adapters."""
    train_dense = False          # any backbone Linear of the block trainable (--fine_tune_to all)
    qkv = (None, None, None)
    d_o = d_i = d_o2 = None


class _Dense:
    """One Linear of the backbone: compute-dtype copies W [out, in] (NT operand) and W^T (dgrad operand).  Frozen: filled once.
    Trainable (--fine_tune_to all, Pretraining/): the copies are re-packed from the flat fp32 master every step and g_w / g_b
    receive dW = dY^T X (a4r_gemm_tn) and db = column sums of dY (a4r_colsum)."""

    def __init__(self, eng, weight, bias, dt, w_dst=None, wT_dst=None, b_dst=None, view2d=None, pad=None):
        out_f, in_f = view2d if view2d is not None else weight.shape          # view2d: a Conv2d weight seen as [out, C*kh*kw]
        op, ip = pad if pad is not None else (out_f, in_f)                     # pad: zero-padded storage (K-Adapter blocks 16 -> 64 wide)
        self.view2d, self.out_f, self.in_f = view2d, out_f, in_f
        self.w = w_dst if w_dst is not None else torch.zeros(op, ip, dtype=dt, device=eng.dev)
        self.wT = wT_dst if wT_dst is not None else torch.zeros(ip, op, dtype=dt, device=eng.dev)
        padded = tuple(self.w.shape) != (out_f, in_f)
        if weight.requires_grad:
            eng.add_pack(weight, self.w, False)
            eng.add_pack(weight, self.wT, True)
        else:
            w2 = weight.detach().reshape(out_f, in_f)
            self.w[:out_f, :in_f].copy_(w2.to(dt))
            self.wT[:in_f, :out_f].copy_(w2.t().to(dt))
        self.g_w = eng.grad_view(weight)
        self.s_w = eng.scratch(*self.w.shape) if (padded and self.g_w is not None) else None
        if self.s_w is not None:
            eng.add_corner(self.s_w, weight, out_f, in_f)
        self.b = self.g_b = self.s_b = None
        if bias is not None:
            if b_dst is None and padded:
                b_dst = torch.zeros(self.w.shape[0], dtype=torch.float32, device=eng.dev)
            if b_dst is not None:
                self.b = b_dst
                if bias.requires_grad:
                    eng.add_pack_bias(bias, b_dst[:out_f])
                else:
                    b_dst[:out_f].copy_(bias.detach().float())
            else:
                self.b = bias.data if bias.requires_grad else eng._f32(bias)     # trainable: the fp32 master (a flat_p view) itself
            self.g_b = eng.grad_view(bias)
            if padded and self.g_b is not None:
                self.s_b = eng.scratch(1, self.w.shape[0])[0]
                eng.add_corner(self.s_b.view(1, -1), bias, 1, out_f)
        self.trainable = self.g_w is not None or self.g_b is not None


class _KAdapter:
    """Engine side of one KAdapterBlock (modules.py:161-206): down (trainable Linear) -> two plain post-LN blocks (all weights
    trainable, no mask, not causal) -> up, + input."""
    pass


class TransRecEngine:
    def __init__(self, model, args, arch='sasrec', dtype='bf16', phm_owner=None):
        self.model, self.args, self.arch = model, args, arch
        self.root = phm_owner if phm_owner is not None else model
        p0 = next(model.parameters())
        self._require_device(p0)
        self.dev = p0.device
        if dtype not in ('bf16', 'fp32', 'fp8'):
            raise ValueError("compute_dtype must be 'bf16', 'fp32' or 'fp8'")
        # fp8: bf16 storage everywhere + OCP e4m3 operands (per-token / per-output-channel scales) for the frozen backbone's forward
        # GEMMs whose input is a LayerNorm output (qkv, FFN-up); everything else, and all of backward, is the bf16 path
        self.fp8 = dtype == 'fp8'
        # bf16 storage: the saved GELU derivative of the encoder FFNs is kept as 8-bit fixed point (include/a4r.h c2_mode 2)
        self.q8_deriv = dtype != 'fp32' and _os.environ.get('A4R_Q8_DERIV', '1') != '0'
        self.T = torch.float32 if dtype == 'fp32' else torch.bfloat16
        # --residual_dtype fp32 (bf16 storage): fp32 twins of the residual stream, keyed by the bf16 tensor they shadow (see _sub_forward)
        self.res32 = dtype != 'fp32' and getattr(args, 'residual_dtype', 'bf20') == 'fp32'
        # --residual_dtype bf24 (round 6): the same twins as ONE byte per element (the 24-bit residual stream of a4r_adapter_ln_fwd, w_frag bit 1) on the
        # sub-layers that run the one-launch serial adapter kernel; the others keep the bf16 stream
        # the residual stream as the bf16 tensor + a low-bit plane: 8 more mantissa bits per element (bf24) or 4 (bf20: half the plane bytes)
        self.res24 = dtype != 'fp32' and getattr(args, 'residual_dtype', 'bf20') in ('bf24', 'bf20')
        self.lo_div = 2 if getattr(args, 'residual_dtype', 'bf20') == 'bf20' else 1               # plane bytes per row = H / lo_div
        self._twin = {}
        # --news_attributes (encoders.py:62-99): rows are [ids | mask] per attribute, laid out title, abstract, body; every attribute runs through the
        # same tower and the item vector is the mean.  With more than one, the attributes are stacked as extra items at the longest length
        # (_stack_attrs): self.S = that length, key masks do the rest.
        na = [a for a in ('title', 'abstract', 'body') if a in set(getattr(args, 'news_attributes', ['title']) or ['title'])] or ['title']
        self.attrs, st = [], 0
        for a in ('title', 'abstract', 'body'):
            nw = int(getattr(args, 'num_words_' + a, 0)) if a in na else 0
            if nw:
                self.attrs.append((st, nw))
            st += 2 * nw
        self.n_attr = len(self.attrs) if type(self).__name__ == 'TransRecEngine' else 1
        self.S = max(nw for _, nw in self.attrs) if self.attrs else getattr(args, 'num_words_title', 0)      # (a single attribute: its own length)
        self.S0 = self.S                           # the title length of the data; self.S may be shorter for one training step (train_forward)
        self.E = args.embedding_dim
        self.Lseq = args.max_seq_len + 1
        self.seed = int(getattr(args, 'dropout_seed', 0x5eed))
        self.step_count = 0
        self._packs_T, self._packs_b, self._virtual = [], [], []
        self._arena, self._arena_used, self._corners, self._corner_tab = [], 0, [], None
        # one-launch adapter + residual + LayerNorm kernels (a4r_adapter_fused.hip); A4R_FUSE_ADAPTERS=0: the three-launch forms (A/B runs, tests)
        self.fuse_adapters = bool(int(_os.environ.get('A4R_FUSE_ADAPTERS', '1'))) and bool(getattr(args, 'fuse_adapters', True))
        self._collect_trainables()
        for p in self.trainable_params:            # resumed run (FusedAdam.load_state_dict): the counter-based dropout stream continues
            if getattr(p, '_a4r_resume_step', None) is not None:
                self.step_count = int(p._a4r_resume_step)
                p._a4r_resume_step = None
        self._build_item_tower()
        self._build_sasrec()
        self._check_coverage()
        self._finalize_packs()
        if self.fp8:
            self._build_fp8()
        self.cap_items = 0
        self.cap_users = 0
        self._bufs, self._saved_bert, self._saved_sas = {}, None, None
        self._fused_opt = None      # FusedAdam once it is bound: p.grad are views of flat_g (see backward_bound)
        self._flat_clean = False    # flat_g was zeroed by optimizer.zero_grad() and nothing has been accumulated since
        self._dirty = {}            # (_buf key) -> rows a partial-row producer has written (see _buf_tail0)
        self._ctx = None
        self._wstream, self._wdone, self._wev = None, None, None     # optional side stream for the adapter weight gradients (A4R_WGRAD_STREAM)
        self._saved_M = self._saved_Mu = 0

    @classmethod
    def inference_snapshot(cls, model, args, arch, dtype, phm_owner=None):
        """A forward-only engine over the CURRENT parameter values in another compute dtype (the fp32 item sweep of eval,
        --eval_compute_dtype): every tensor is packed once as frozen, nothing is re-viewed, so the training engine that owns the
        parameters' flat buffer is not disturbed.  Build, use, drop (the copies go stale at the next optimizer step)."""
        root = phm_owner if phm_owner is not None else model
        flags = [(p, p.requires_grad) for p in root.parameters()]
        try:
            for p, _ in flags:
                p.requires_grad_(False)
            return cls(model, args, arch=arch, dtype=dtype, phm_owner=phm_owner)
        finally:
            for p, f in flags:
                p.requires_grad_(f)

    FP8_U_SCALE = 0.25          # static scale of the FFN's GELU output as e4m3 (|u| <= 112 representable, subnormal step 5e-4)
    FP8_DU_MARGIN = 16.0        # c_scale of the d FFN-down output = margin x max column norm of W2 (see _build_fp8)

    def _build_fp8(self):
        """e4m3 copies (+ per-output-channel scales) of the FROZEN operands of every item-tower block: forward qkv, attention-output,
        FFN-up, FFN-down, and the two FFN dgrad operands (W2^T for du = (d_o W2) * gelu', W1^T for dn2 = du W1).
        The dgrad chain keeps ONE scale per token row: d_o is quantised per row (absmax / 448); du inherits that row scale times
        c_du = FP8_DU_MARGIN x max_c |W2[:, c]|_2 -- |du[m, c]| <= |d_o[m]|_2 |W2[:, c]|_2 1.13 <= 31 x absmax(d_o[m]) |W2[:, c]|_2 at the very
        worst (all elements equal and aligned), ~1.2 x typically: with the margin at 16 a typical element is stored near 8 (of 448) and e4m3,
        being floating point, loses no precision to the head-room."""
        # (pre-LN image tower: engine_vit.py -- both fp8 GEMM inputs are LayerNorm outputs; post-LN text tower: _block_forward below -- the
        # block input and the attention-output sub-layer's LayerNorm output leave the fused adapter kernel as e4m3 + row scale)
        ok = lambda w: w.shape[0] % 256 == 0 and w.shape[1] % 128 == 0
        more = _os.environ.get('A4R_FP8_MORE', '1') != '0'              # 0: round 2's coverage (forward qkv + FFN-up only; A/B runs)
        for b in self.bert_blocks:
            b.wqkv8 = b.wi8 = b.wo8 = b.wo28 = b.wo2T8 = b.wiT8 = None
            b.wqkv8_dyn = False
            frozen_qkv = not b.lora and all(d is not None and not d.trainable for d in b.qkv)
            if frozen_qkv and ok(b.wqkv):
                b.wqkv8, b.wqkv8s = L.quantize_weight_fp8(b.wqkv)
            elif more and b.lora and all(d is None or not d.trainable for d in b.qkv) and ok(b.wqkv):       # (None: a LoRA-carrying slot)
                # LoRA on q / v (configs[2]): the FORWARD operand is the merged W + B A / r, re-quantised after every merge (pack_trainables:
                # one row pass over [3H, H] per layer); the LoRA gradients and dx keep using the bf16 operands
                b.wqkv8 = torch.zeros(b.wqkv.shape, dtype=torch.uint8, device=b.wqkv.device)
                b.wqkv8s = torch.zeros(b.wqkv.shape[0], dtype=torch.float32, device=b.wqkv.device)
                b.wqkv8_dyn = True
            if not b.d_i.trainable and ok(b.wi):
                b.wi8, b.wi8s = L.quantize_weight_fp8(b.wi)
            if more and not b.d_o.trainable and ok(b.wo):
                b.wo8, b.wo8s = L.quantize_weight_fp8(b.wo)
            if more and b.wi8 is not None and not b.d_o2.trainable and ok(b.wo2) and ok(b.wo2T) and ok(b.wiT) and self._q8(b):
                b.wo28, b.wo28s = L.quantize_weight_fp8(b.wo2)          # forward FFN-down  [H, F]
                b.wo2T8, b.wo2T8s = L.quantize_weight_fp8(b.wo2T)       # d FFN-down        [F, H]: rows = columns of du
                b.wiT8, b.wiT8s = L.quantize_weight_fp8(b.wiT)          # d FFN-up          [H, F]
                b.c_du = float(self.FP8_DU_MARGIN * b.wo2T.float().norm(dim=1).max())

    def _const_rows(self, name, rows, value):
        """[rows, 1] fp32 buffer filled with `value` (the constant scale_a of an fp8 GEMM whose A operand carries a static scale)."""
        key = ('const.' + name, 1, torch.float32)
        t = self._bufs.get(key)
        if t is None or t.shape[0] < rows:
            t = torch.full((rows, 1), float(value), dtype=torch.float32, device=self.dev)
            self._bufs[key] = t
        return t[:rows]

    def _require_device(self, p0):
        if not p0.is_cuda:
            raise RuntimeError('adapter4rec_amd runs on an MI355X only: move the model to a cuda device first '
                               '(there is no CPU / PyTorch fallback path)')
        L.lib()

    # ------------------------------------------------------------------ trainable bookkeeping
    def _collect_trainables(self):
        named = [(n, p) for n, p in self.root.named_parameters() if p.requires_grad]
        self.trainable_names = [n for n, _ in named]
        self.trainable_params = [p for _, p in named]
        self.n_trainable = len(named)
        sizes = [p.numel() for p in self.trainable_params]
        self.offsets, o = {}, 0
        for p, n in zip(self.trainable_params, sizes):
            o = pad_to(o, 4)                                  # 16-byte aligned segments (row kernels load gamma/beta as uint4)
            self.offsets[id(p)] = (o, n)
            o += n
        total = pad_to(max(o, 4), 4)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_gs = torch.zeros(total, dtype=torch.float32, device=self.dev)      # scratch target of the autograd path
        self._grad_target = self.flat_gs
        for p in self.trainable_params:                       # parameters become views into the flat buffer
            off, n = self.offsets[id(p)]
            if p.dtype != torch.float32:
                raise TypeError('trainable parameters must be fp32 (master copy)')
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            p._a4r_flat = (self, off, n)
        self._covered = set()

    def grad_view(self, p):
        """fp32 view for p's gradient inside the CURRENT target flat buffer, resolved at use time."""
        if not p.requires_grad:
            return None
        self._covered.add(id(p))
        off, n = self.offsets[id(p)]
        shape = tuple(p.shape)
        return lambda: self._grad_target[off:off + n].view(shape)

    ARENA = 4 << 20          # fp32 elements per chunk of the gradient-scratch arena (16 MiB)

    def scratch(self, rows, cols):
        """Zero-padded fp32 scratch matrix for a weight-gradient GEMM whose true shape is not a tile multiple.  All of them live in a
        few large chunks: ONE memset per chunk clears them at the start of backward, ONE a4r_unpack_add launch at its end adds the
        valid corners into the flat gradient (they used to be a zero_() and an add_() per matrix per step)."""
        n = pad_to(rows * cols, 4)
        if not self._arena or self._arena_used + n > self._arena[-1].numel():
            self._arena.append(torch.zeros(max(self.ARENA, n), dtype=torch.float32, device=self.dev))
            self._arena_used = 0
        t = self._arena[-1][self._arena_used:self._arena_used + rows * cols].view(rows, cols)
        self._arena_used += n
        return t

    def add_corner(self, src, p, rows, cols, alpha=1.0):
        if p.requires_grad:
            self._corners.append((src, p, rows, cols, float(alpha)))

    def _zero_scratch(self):
        for c in self._arena:
            L.zero(c)

    def _flush_corners(self):
        if not self._corners:
            return
        if self._corner_tab is None:
            ents = [L.AddDesc(src.data_ptr(), self.offsets[id(p)][0], rows, cols, src.stride(0), alpha) for src, p, rows, cols, alpha in self._corners]
            self._corner_tab = (L.desc_table(ents, self.dev), len(ents), max(r * c for _, _, r, c, _ in self._corners))
        tab, n, mx = self._corner_tab
        L.unpack_add(self._grad_target, tab, n, mx)

    def add_pack(self, p, dst, transpose):
        self._packs_T.append((p, dst, transpose))

    def add_pack_bias(self, p, dst):
        self._packs_b.append((p, dst))

    def add_virtual(self, adapter):
        self._virtual.append(adapter)

    def _check_coverage(self):
        missing = [n for n, p in zip(self.trainable_names, self.trainable_params) if id(p) not in self._covered]
        if missing:
            raise NotImplementedError(
                'the native path trains adapter / Pfeiffer-LN / LayerNorm / compacter tensors (frozen backbone, dgrad only); '
                f'no native weight-gradient kernel for: {missing[:6]}{" ..." if len(missing) > 6 else ""} '
                '(full fine-tuning, --fine_tune_to all, is a later row of SURVEY.md section 8(f))')

    def _finalize_packs(self):
        """Descriptor tables (device) for a4r_pack_matrices: parameters live in flat_p, so src_off is static."""
        def table(entries, frozen_src=None):
            if not entries:
                return None
            arr = (L.PackDesc * len(entries))()
            mx = 0
            for i, (p, dst, tr) in enumerate(entries):
                rows, cols = (p.shape[0], p[0].numel()) if p.dim() >= 2 else (1, p.shape[0])
                rp, cp = (dst.shape[0], dst.shape[1]) if dst.dim() == 2 else (1, dst.shape[0])
                off = self.offsets[id(p)][0] if frozen_src is None else frozen_src[i]
                ld = dst.stride(0) if dst.dim() == 2 and dst.stride(0) != cp else 0      # column block of a fused operand
                arr[i] = L.PackDesc(off, dst.data_ptr(), rows, cols, rp, cp, int(tr), ld)
                mx = max(mx, rp * cp)
            return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev), len(entries), mx
        code = lambda t: L.BF16 if t.dtype == torch.bfloat16 else L.F32
        every = list(self._packs_T) + [(p, d, False) for p, d in self._packs_b]
        # one table per destination dtype (BERT-side copies are in compute dtype, SASRec-side and biases fp32)
        self._tabs = []
        for c in (L.BF16, L.F32):
            ents = [(p, d, t) for p, d, t in every if p.requires_grad and code(d) == c]
            if ents:
                self._tabs.append(table(ents) + (c,))
        # frozen adapters (e.g. eval of a loaded checkpoint with requires_grad False): packed once from a private flat copy
        frozen = [(p, d, t) for p, d, t in every if not p.requires_grad]
        if frozen:
            src = torch.cat([p.data.reshape(-1).float() for p, _, _ in frozen])
            offs, o = [], 0
            for p, _, _ in frozen:
                offs.append(o)
                o += p.numel()
            for c in (L.BF16, L.F32):
                idx = [i for i, (_, d, _) in enumerate(frozen) if code(d) == c]
                if idx:
                    t = table([frozen[i] for i in idx], [offs[i] for i in idx])
                    L.pack_matrices(src, t[0], t[1], t[2], c)
            if self.dev.type == 'cuda':
                torch.cuda.current_stream().synchronize()
        # compacter: effective matrices are functions of (phm_rule, W_left, W_right): built by torch each step
        self._virt_flat, self._tabs_virt = None, []
        if self._virtual:
            n = sum(2 * a.d * a.width for a in self._virtual)
            self._virt_flat = torch.zeros(n, dtype=torch.float32, device=self.dev)
            ents, offs, o = [], [], 0
            for a in self._virtual:
                a.v_off = o
                wd_shape = torch.empty(a.d, a.width)       # [d, H] then [H, d]
                wu_shape = torch.empty(a.width, a.d)
                for shp, dst, tr in ((wd_shape, a.wd, False), (wd_shape, a.wdT, True)):
                    ents.append((shp, dst, tr)); offs.append(o)
                o += a.d * a.width
                for shp, dst, tr in ((wu_shape, a.wu, False), (wu_shape, a.wuT, True)):
                    ents.append((shp, dst, tr)); offs.append(o)
                if a.frag_f is not None:                   # fragment-ordered copies of the effective matrices
                    for shp, dst, tr in a.frag_entries(wd_shape, wu_shape):
                        ents.append((shp, dst, tr)); offs.append(o - a.d * a.width if shp is wd_shape else o)
                o += a.d * a.width
            for c in (L.BF16, L.F32):
                idx = [i for i, (_, d, _) in enumerate(ents) if code(d) == c]
                if idx:
                    self._tabs_virt.append(table([ents[i] for i in idx], [offs[i] for i in idx]) + (c,))
        # a4r_phm_build / a4r_phm_bwd descriptors: two PHMLinear per virtual adapter (down: [d, width] at v_off, up: [width, d] after it)
        self._phm_tab, self._phm_n = None, 0
        if self._virtual:
            leaves = [q for a in self._virtual for lin in a.virtual for q in (lin.phm_rule, lin.W_left, lin.W_right)]
            if all(q.requires_grad for q in leaves):
                off = lambda q: self.offsets[id(q)][0]
                ents = []
                for a in self._virtual:
                    dn, up = a.virtual
                    ents.append(L.PhmDesc(off(dn.phm_rule), off(dn.W_left), off(dn.W_right), a.v_off, a.s_wd.data_ptr(), a.s_wd.stride(0),
                                          dn.in_features, dn.out_features, dn.phm_dim, 0))
                    ents.append(L.PhmDesc(off(up.phm_rule), off(up.W_left), off(up.W_right), a.v_off + a.d * a.width, a.s_wu.data_ptr(),
                                          a.s_wu.stride(0), up.in_features, up.out_features, up.phm_dim, 0))
                self._phm_tab, self._phm_n = L.desc_table(ents, self.dev), len(ents)
            elif any(q.requires_grad for q in leaves):
                raise NotImplementedError('Compacter tensors must be trainable (or frozen) together')
            else:
                # frozen Compacter tensors (a forward-only snapshot engine, e.g. the fp32 evaluation sweep of a trained model): the leaves are not in
                # the flat trainable buffer -- a flat device copy of their own and the same a4r_phm_build launch (no gradient scratch: build only)
                uniq, o = {}, 0
                for q in leaves:
                    if id(q) not in uniq:
                        uniq[id(q)] = (o, q)
                        o += (q.numel() + 3) // 4 * 4
                self._phm_frozen = torch.zeros(max(o, 4), dtype=torch.float32, device=self.dev)
                for o_, q in uniq.values():
                    self._phm_frozen[o_:o_ + q.numel()].copy_(q.detach().reshape(-1))
                off = lambda q: uniq[id(q)][0]
                ents = []
                for a in self._virtual:
                    dn, up = a.virtual
                    ents.append(L.PhmDesc(off(dn.phm_rule), off(dn.W_left), off(dn.W_right), a.v_off, 0, 0, dn.in_features, dn.out_features, dn.phm_dim, 0))
                    ents.append(L.PhmDesc(off(up.phm_rule), off(up.W_left), off(up.W_right), a.v_off + a.d * a.width, 0, 0, up.in_features, up.out_features, up.phm_dim, 0))
                self._phm_frozen_tab = (L.desc_table(ents, self.dev), len(ents))

    def pack_trainables(self):
        """Refresh the kernel-side copies of the trainable matrices (call after every optimiser step)."""
        for tab, n, mx, c in self._tabs:
            L.pack_matrices(self.flat_p, tab, n, mx, c)
        # LoRA: re-merge W + B A / r into the packed qkv operands -- every LoRA-carrying projection of a dtype in one launch (a4r_lora_merge_batch)
        if getattr(self, '_lora_tabs', None) is None:
            groups = {}
            for blk in self.bert_blocks + self.sas_blocks:
                for lo in blk.lora:
                    H, sl, m = blk.H, lo.slot, lo.mod
                    groups.setdefault(blk.wqkv.dtype, []).append(
                        (m.weight.data, m.lora_A.data if lo.r else None, m.lora_B.data if lo.r else None, lo.scaling,
                         blk.wqkv[sl * H:(sl + 1) * H], blk.wqkvT[:, sl * H:(sl + 1) * H], lo.r))
            self._lora_tabs = [L.lora_table(ents, self.dev) for ents in groups.values()]
        for tab in self._lora_tabs:
            L.lora_merge_batch(tab)
        if self.fp8:
            for blk in self.bert_blocks:
                if getattr(blk, 'wqkv8_dyn', False):
                    L.quant_rows_fp8(blk.wqkv, blk.wqkv8, blk.wqkv8s.view(-1, 1))
        if self._virtual:
            if self._phm_tab is not None:                     # Compacter: effective matrices from (phm_rule, W_left, W_right), one launch
                L.phm_build(self.flat_p, self._phm_tab, self._phm_n, self._virt_flat)
            else:                                             # frozen Compacter tensors (inference snapshot): their own flat copy, the same kernel
                L.phm_build(self._phm_frozen, self._phm_frozen_tab[0], self._phm_frozen_tab[1], self._virt_flat)
            for tab, n, mx, c in self._tabs_virt:
                L.pack_matrices(self._virt_flat, tab, n, mx, c)

    # ------------------------------------------------------------------ frozen weight packing
    def _w(self, t, dt=None):
        return t.detach().to(self.dev, dt or self.T).contiguous()

    def _wT(self, t, dt=None):
        return t.detach().to(self.dev, dt or self.T).t().contiguous()

    def _f32(self, t):
        return t.detach().to(self.dev, torch.float32).contiguous()

    def _adapter_of(self, wrapper, attr, width, dt):
        mod = getattr(wrapper, attr, None)
        if mod is None:
            return None
        if not isinstance(mod, (AdapterBlock, AdapterPfeifferBlock, HyperComplexAdapterBlock)):
            raise NotImplementedError(f'adapter module {type(mod).__name__}')
        return _Adapter(mod, width, self, dt)

    def _so(self, mod, width, dt):
        """(dense, LayerNorm, adapter, placement, LN_new) of a plain or wrapped self-output."""
        if hasattr(mod, 'self_output'):
            so = mod.self_output
            placement = getattr(mod, 'placement', 'serial')
            ln_new = _LN(mod.LN, self) if hasattr(mod, 'LN') else None
            return so.dense, so.LayerNorm, self._adapter_of(mod, 'adapter', width, dt), placement, ln_new
        return mod.dense, mod.LayerNorm, None, None, None

    def _build_item_tower(self):
        bert = self.model.bert_encoder.text_encoders['title'].bert_model
        kmod = None
        if type(bert).__name__ == 'BertKAdaptedBertModel':       # K-Adapter wraps the backbone (model.py:523-559)
            kmod, bert = bert, bert.bert_model
        self.geo = backbone_geometry(bert)
        if self.geo['hidden_act'] not in ('gelu',):
            raise NotImplementedError(f"hidden_act {self.geo['hidden_act']}")
        self._build_bert(bert)
        self._build_head()
        self.bert_trains = any(p.requires_grad for p in bert.parameters())
        self.bert_kads, self.bert_klist, self.d_com = [], [], None
        if kmod is not None:
            nb = len(self.bert_blocks)
            self.bert_klist = [int(k) for k in kmod.k_adapter_num_list]
            if any(k < 1 or k > nb for k in self.bert_klist):
                raise ValueError(f'--k_adapter_bert_list {self.bert_klist} outside 1..{nb}')
            self.bert_kads = [self._make_kadapter(a, self.H, self.S, self.T, 6000 + 64 * j) for j, a in enumerate(kmod.bert_adapter_list)]
            self.d_com = _Dense(self, kmod.com_dense.weight, kmod.com_dense.bias, self.T)
            self.cls_only = False                      # the adapters attend over all tokens of the last layer's output

    def _build_bert(self, bert):
        g = self.geo
        H, nh = g['hidden_size'], g['num_attention_heads']
        self.H, self.F = H, g['intermediate_size']
        # (--num_words_title > 32, round 5: the long attention kernels with the titles' key mask, head width 64 -- every BERT size of run.py:100-114)
        if H % 64 or self.F % 64 or (H // nh) not in (32, 64) or self.S > 256 or (self.S > 32 and H // nh != 64):
            raise NotImplementedError(f'encoder geometry H={H} F={self.F} heads={nh} S={self.S}')
        emb = bert.embeddings
        tab = lambda p: p.data if p.requires_grad else self._f32(p)        # trainable tables are read from the flat fp32 master
        wemb = emb.word_embeddings
        self.prompt_n, self.g_prompt = 0, None
        if type(wemb).__name__ == 'SoftEmbedding':      # soft prompt (model.py:586-630): rows V .. V+n-1 of an extended table hold the
            self.prompt_n = int(wemb.n_tokens)          # learned vectors and the first n ids of every title are redirected to them
            if self.prompt_n > self.S:
                raise ValueError(f'--n_tokens {self.prompt_n} exceeds the title length {self.S}')
            self.prompt_param = wemb.learned_embedding
            self.g_prompt = self.grad_view(wemb.learned_embedding)
            wemb = wemb.wte
            if wemb.weight.requires_grad:
                raise NotImplementedError('soft prompt together with a trainable vocabulary table')
            V = wemb.weight.shape[0]
            self.prompt_V = V
            self.emb_word_ext = torch.cat([self._f32(wemb.weight), torch.zeros(self.prompt_n, H, device=self.dev)], 0)
        self.emb_word, self.emb_pos = tab(wemb.weight), tab(emb.position_embeddings.weight)
        self.emb_type = tab(emb.token_type_embeddings.weight)
        self.emb_type0 = self.emb_type[0]
        self.emb_ln = _LN(emb.LayerNorm, self)
        self.g_word, self.g_pos = self.grad_view(wemb.weight), self.grad_view(emb.position_embeddings.weight)
        self.g_type = self.grad_view(emb.token_type_embeddings.weight)
        # --finetune_layernorm (run.py:496-501) / --fine_tune_to all / soft prompt: the embedding side needs its input gradient
        self.train_emb = any(f is not None for f in (self.g_word, self.g_pos, self.g_type, self.emb_ln.g_gamma, self.emb_ln.g_beta, self.g_prompt))
        if getattr(bert, 'pooler', None) is not None:                       # never on the forward path ([0][:, 0], encoders.py:55): zero gradient
            for p in bert.pooler.parameters():
                self.grad_view(p)
        self.cls_only = bool(getattr(self.args, 'cls_only_last', True))
        self.roberta = g['model_type'] == 'roberta'
        self.pad_id = int(g['pad_token_id'])
        self.p_hidden = float(g['hidden_dropout_prob'])
        self.p_attn = float(g['attention_probs_dropout_prob'])
        self.bert_blocks = []
        for i, layer in enumerate(bert.encoder.layer):
            b = _Block()
            att = layer.attention.self
            for lin in (att.query, att.key, att.value):
                if type(lin).__name__ not in ('LoRALinear', 'Linear'):
                    raise NotImplementedError(f'projection module {type(lin).__name__}')
            b.lora = _Lora.for_block((att.query, att.key, att.value), H, self, self.T)
            b.H, b.F, b.nh, b.dh, b.S = H, self.F, nh, H // nh, self.S
            b.is_item = True                       # (a block of the item tower: follows self._pk)
            b.long = self.S > 32                   # titles of more than 32 tokens: a4r_attn_long_* with the key mask (any step length <= S)
            b.causal, b.mask_neg, b.scale = False, FMIN, 1.0 / math.sqrt(H // nh)
            b.ffn_act = L.ACT_GELU
            b.p_hidden, b.p_attn, b.site = self.p_hidden, self.p_attn, 16 * i
            b.wqkv = torch.zeros(3 * H, H, dtype=self.T, device=self.dev)
            b.wqkvT = torch.zeros(H, 3 * H, dtype=self.T, device=self.dev)
            b.bqkv = torch.zeros(3 * H, dtype=torch.float32, device=self.dev)
            self._pack_lora_bias(b, H)
            b.qkv = tuple(None if type(lin).__name__ == 'LoRALinear' else          # LoRA slots are merged in pack_trainables
                          _Dense(self, lin.weight, lin.bias, self.T, b.wqkv[sl * H:(sl + 1) * H], b.wqkvT[:, sl * H:(sl + 1) * H],
                                 b.bqkv[sl * H:(sl + 1) * H])
                          for sl, lin in enumerate((att.query, att.key, att.value)))
            d1, ln1, ad1, pl1, lnn1 = self._so(layer.attention.output, H, self.T)
            d2, ln2, ad2, pl2, lnn2 = self._so(layer.output, H, self.T)
            b.d_o = _Dense(self, d1.weight, d1.bias, self.T)
            b.d_i = _Dense(self, layer.intermediate.dense.weight, layer.intermediate.dense.bias, self.T)
            b.d_o2 = _Dense(self, d2.weight, d2.bias, self.T)
            b.wo, b.woT, b.bo = b.d_o.w, b.d_o.wT, b.d_o.b
            b.wi, b.wiT, b.bi = b.d_i.w, b.d_i.wT, b.d_i.b
            b.wo2, b.wo2T, b.bo2 = b.d_o2.w, b.d_o2.wT, b.d_o2.b
            b.train_dense = any(d is not None and d.trainable for d in b.qkv + (b.d_o, b.d_i, b.d_o2))
            b.ln1, b.ln2 = _LN(ln1, self), _LN(ln2, self)
            b.ad1, b.ad2, b.pl1, b.pl2, b.lnn1, b.lnn2 = ad1, ad2, pl1, pl2, lnn1, lnn2
            b.need_dx = i > 0 or self.train_emb
            b.T = self.T
            self.bert_blocks.append(b)
        lastb = self.bert_blocks[-1]
        if lastb.pl1 == 'parallel' or lastb.pl2 == 'parallel':
            self.cls_only = False                  # the parallel form needs the sub-layer inputs per token row

    def _build_head(self):
        fc = self.model.bert_encoder.text_encoders['title'].fc
        self.d_fc = _Dense(self, fc.weight, fc.bias, self.T)
        self.fc_w, self.fc_b = self.d_fc.w, self.d_fc.b        # [E, H] compute dtype (forward operand)
        self.fc_wT32 = torch.zeros(fc.in_features, fc.out_features, dtype=torch.float32, device=self.dev)   # [H, E] fp32 (dgrad operand: d_pre is fp32)
        if fc.weight.requires_grad:
            self.add_pack(fc.weight, self.fc_wT32, True)
        else:
            self.fc_wT32.copy_(fc.weight.detach().t().float())
        if self.E % 64:
            raise NotImplementedError('item head: embedding_dim % 64 == 0')

    def _build_sasrec(self):
        te = self.model.user_encoder.transformer_encoder
        E, nh = self.E, self.args.num_attention_heads
        # (head widths 128 / 256: --embedding_dim 256 / 512 with the default two heads, parameters.py:27-28 -- fp32 instantiations of the short attention kernel)
        # (histories of more than 32 items, --max_seq_len > 32: the causal, key-masked form of the long attention kernels, head width 32 / 64; 128 up to 128 positions)
        if (E // nh) not in (32, 64, 128, 256) or E % nh or self.Lseq - 1 > 256 or (self.Lseq - 1 > 32 and (E // nh) not in (32, 64)
                                                                                   and not (E // nh == 128 and self.Lseq - 1 <= 128)):
            raise NotImplementedError(f'SASRec geometry E={E} heads={nh} T={self.Lseq - 1}')
        pe = te.position_embedding.weight
        self.pos_emb = pe.data if pe.requires_grad else self._f32(pe)
        self.g_pos_emb = self.grad_view(pe)
        self.sas_ln0 = _LN(te.layer_norm, self)
        self.p_sas = float(self.args.drop_rate)
        f32 = torch.float32
        self.sas_blocks = []
        kad = type(te.transformer_blocks).__name__ == 'SASRecKAdaptedTransformerBlocks'       # model.py:562-583
        block_list = te.transformer_blocks.transformer_blocks if kad else te.transformer_blocks
        for j, blk in enumerate(block_list):
            tb = blk.transformer_block if hasattr(blk, 'transformer_block') else blk
            b = self._make_block(tb, E, nh, self.Lseq - 1, f32, True, -1e9, self.p_sas, 4096 + 16 * j)
            placement = getattr(blk, 'placement', None) if blk is not tb else None
            if blk is not tb:
                if placement == 'pfeiffer':
                    b.ad2, b.pl2, b.lnn2 = self._adapter_of(blk, 'adapter', E, f32), 'pfeiffer', _LN(blk.LN, self)
                else:
                    b.ad1 = self._adapter_of(blk, 'adapter1', E, f32)
                    b.ad2 = self._adapter_of(blk, 'adapter2', E, f32)
                    b.pl1 = (placement or 'serial') if b.ad1 else None
                    b.pl2 = (placement or 'serial') if b.ad2 else None
            self.sas_blocks.append(b)
        self.sas_kads, self.d_com2 = [], None
        if kad:
            tbs = te.transformer_blocks
            self.sas_kads = [self._make_kadapter(m, E, self.Lseq - 1, f32, 5000 + 64 * j) for j, m in enumerate(tbs.adapter_list)]
            self.d_com2 = _Dense(self, tbs.com_dense2.weight, tbs.com_dense2.bias, f32)

    SAS_FUSED = bool(int(_os.environ.get('A4R_SAS_FUSED', '1')))       # 0: the multi-launch user tower (A/B runs)

    def _sas_fused_ok(self):
        """The one-launch-per-block kernels (a4r_sasrec.hip) serve the user tower when every block is the reference's default shape:
        64 wide, 2 heads x 32, d_inner 256, frozen dense weights and LayerNorms, a serial Houlsby or Compacter adapter after both
        sub-layers with bottleneck <= 32, no LoRA / Pfeiffer / parallel / K-Adapter.  Anything else keeps the multi-launch path."""
        if not self.SAS_FUSED or self.sas_kads or not self.sas_blocks or self.Lseq - 1 > 32:
            return False
        for b in self.sas_blocks:
            if (b.H, getattr(b, 'Hv', b.H), b.nh, b.F, b.T) != (64, 64, 2, 256, torch.float32) or not b.causal or b.lora or b.train_dense:
                return False
            pfe = b.ad1 is None and b.ad2 is not None and b.pl2 == 'pfeiffer' and b.lnn2 is not None       # SASRecPfeifferAdaptedSelfOutput
            ser = b.ad1 is not None and b.ad2 is not None and b.pl1 == 'serial' and b.pl2 == 'serial' and b.lnn1 is None and b.lnn2 is None
            if not (pfe or ser):
                return False
            ads = (b.ad2,) if pfe else (b.ad1, b.ad2)
            if any(a.d > 32 or a.dp != 64 or a.act != b.ad2.act or a.kind != b.ad2.kind or a.d != b.ad2.d or a.virtual is not None and pfe for a in ads):
                return False
            if any(f is not None for ln in (b.ln1, b.ln2) for f in (ln.g_gamma, ln.g_beta)):
                return False
        return True

    def _sas_desc(self, b, seed, with_grads):
        """a4r_sasrec_block_t fields of one block (the gradient sinks are resolved against the CURRENT gradient target)."""
        gg = lambda f: f() if f is not None else None
        a0 = b.ad1 if b.ad1 is not None else b.ad2
        d = dict(wqkv=b.wqkv, wfc=b.wo, w1=b.wi, b1=b.bi, w2=b.wo2, b2=b.bo2, ln1_g=b.ln1.gamma, ln1_b=b.ln1.beta, ln2_g=b.ln2.gamma, ln2_b=b.ln2.beta,
                 E=64, n_heads=2, F=256, d=a0.d, ldwu=a0.dp, ldg_d=64, ldg_u=64, act=a0.act, inner_res=int(a0.kind == 'houlsby'), mode=0,
                 eps=float(b.ln1.eps), mask_neg=float(b.mask_neg), drop_attn=float(b.p_attn), drop_hidden=float(b.p_hidden),
                 drop_site=int(b.site), drop_seed=int(seed))
        pfe = b.ad1 is None
        if pfe:                                     # mode 1: adapter 1's operand slots are unused (non-null for the argument check)
            d.update(mode=1, ln3_g=b.lnn2.gamma, ln3_b=b.lnn2.beta, g_ln3_g=gg(b.lnn2.g_gamma) if with_grads else None,
                     g_ln3_b=gg(b.lnn2.g_beta) if with_grads else None, inner_res=0, d=b.ad2.d, ldwu=b.ad2.dp, act=b.ad2.act)
        for k, ad in (('1', b.ad2 if pfe else b.ad1), ('2', b.ad2)):
            d.update({'wd' + k: ad.wd, 'bd' + k: ad.bd, 'wu' + k: ad.wu, 'bu' + k: ad.bu})
            trains = with_grads and (ad.virtual is not None or ad.g_wu is not None) and not (pfe and k == '1')
            # (d < 64: the zero-padded scratch matrices whose valid corners _flush_corners / a4r_phm_bwd pick up, as the multi-launch path)
            wg = with_grads and not (pfe and k == '1')
            d.update({'g_wd' + k: ad.s_wd if trains else None, 'g_wu' + k: ad.s_wu if trains else None,
                      'g_bd' + k: (ad.s_bd if ad.s_bd is not None else gg(ad.g_bd)) if (wg and ad.g_bd is not None) else None,
                      'g_bu' + k: gg(ad.g_bu) if wg else None})
        return d

    def _make_block(self, tb, Hv, nh, S, dt, causal, mask_neg, p_drop, site):
        """A plain post-LN TransformerBlock (modules.py:16-87: bias-free w_Q / w_K / w_V / fc, ReLU FFN with biases, LayerNorm eps
        1e-6) as an engine _Block.  Widths that are not multiples of 64 (K-Adapter blocks: 16) are stored zero-padded to 64; the
        LayerNorms run on the valid columns only (b.Hv)."""
        mha, ff = tb.multi_head_attention, tb.feed_forward
        H = pad_to(Hv, 64)
        Fv = ff.w_1.out_features
        F = pad_to(Fv, 64)
        dh = Hv // nh
        long = S > 32                 # K-Adapter blocks over the ViT token rows (S = 197 / 50): a4r_attn_long_*, no mask, dh 64 / 32
        wide = dh in (128, 256) and dt == torch.float32 and not long          # (the user tower at --embedding_dim 256 / 512)
        wide_long = long and dh == 128 and dt == torch.float32 and S <= 128      # (--embedding_dim 256 with two heads AND --max_seq_len 33 .. 128: the long kernels' fp32 head width 128)
        if dh not in (32, 64) and not wide and not wide_long and not (0 < dh <= 16 and not long) or S > 256 or (long and H != Hv):
            raise NotImplementedError(f'transformer block geometry width={Hv} heads={nh} S={S}')
        b = _Block()
        b.long = long
        for lin in (mha.w_Q, mha.w_K, mha.w_V):
            if type(lin).__name__ not in ('LoRALinear', 'Linear'):
                raise NotImplementedError(f'projection module {type(lin).__name__}')
        b.lora = _Lora.for_block((mha.w_Q, mha.w_K, mha.w_V), Hv, self, dt)
        if b.lora and H != Hv:
            raise NotImplementedError('LoRA on a zero-padded block')
        b.H, b.Hv, b.F, b.nh, b.dh, b.S = H, Hv, F, nh, dh, S
        b.causal, b.mask_neg, b.scale = causal, mask_neg, 1.0 / math.sqrt(dh)
        b.ffn_act = L.ACT_RELU
        b.p_hidden, b.p_attn, b.site = p_drop, p_drop, site
        b.wqkv = torch.zeros(3 * H, H, dtype=dt, device=self.dev)
        b.wqkvT = torch.zeros(H, 3 * H, dtype=dt, device=self.dev)
        b.bqkv = torch.zeros(3 * H, dtype=torch.float32, device=self.dev) if b.lora else None       # lora.Linear carries a bias
        self._pack_lora_bias(b, H)
        b.qkv = tuple(None if type(lin).__name__ == 'LoRALinear' else
                      _Dense(self, lin.weight, None, dt, b.wqkv[sl * H:(sl + 1) * H], b.wqkvT[:, sl * H:(sl + 1) * H])
                      for sl, lin in enumerate((mha.w_Q, mha.w_K, mha.w_V)))
        b.d_o = _Dense(self, mha.fc.weight, None, dt, pad=(H, H))
        b.d_i = _Dense(self, ff.w_1.weight, ff.w_1.bias, dt, pad=(F, H))
        b.d_o2 = _Dense(self, ff.w_2.weight, ff.w_2.bias, dt, pad=(H, F))
        b.wo, b.woT, b.bo = b.d_o.w, b.d_o.wT, None
        b.wi, b.wiT, b.bi = b.d_i.w, b.d_i.wT, b.d_i.b
        b.wo2, b.wo2T, b.bo2 = b.d_o2.w, b.d_o2.wT, b.d_o2.b
        b.train_dense = any(d is not None and d.trainable for d in b.qkv + (b.d_o, b.d_i, b.d_o2))
        b.ln1, b.ln2 = _LN(mha.layer_norm, self), _LN(ff.layer_norm, self)
        b.ad1 = b.ad2 = b.lnn1 = b.lnn2 = None
        b.pl1 = b.pl2 = None
        b.need_dx = True
        b.T = dt
        return b

    def _pack_lora_bias(self, b, H):
        """lora.Linear keeps a (trainable) bias: its fp32 master is copied into the fused qkv bias by the per-step pack launch."""
        for lo in b.lora:
            if lo.mod.bias is not None:
                self.add_pack_bias(lo.mod.bias, b.bqkv[lo.slot * H:(lo.slot + 1) * H])

    def _make_kadapter(self, mod, width, S, dt, site):
        k = _KAdapter()
        d = mod.down_project.out_features
        dp = pad_to(d, 64)
        k.width, k.d, k.dp, k.T = width, d, dp, dt
        k.down = _Dense(self, mod.down_project.weight, mod.down_project.bias, dt, pad=(dp, width))
        k.up = _Dense(self, mod.up_project.weight, mod.up_project.bias, dt, pad=(width, dp))
        p_drop = float(mod.transformer_blocks[0].multi_head_attention.dropout.p)
        k.blocks = [self._make_block(tb, d, mod.num_head, S, dt, False, 0.0, p_drop, site + 16 * j)
                    for j, tb in enumerate(mod.transformer_blocks)]
        k.tag = f'kad{site}'
        return k

    def _kad_forward(self, k, fus, n_seq, M, train, seed, out, saved):
        """out = fus + up(block2(block1(down(fus))));  saved: per-adapter buffer dict (training) or None."""
        T, dp = k.T, k.dp
        dn = self._buf(k.tag + '.dn', M, dp, T)
        L.gemm_nt(fus, k.down.w, dn, bias=k.down.b, M=M)
        x = dn
        for j, blk in enumerate(k.blocks):
            bufs = saved['blk'][j] if saved is not None else self._block_bufs(k.tag + f'.b{j}.shared', blk, M, True)
            nxt = (saved['bo'] if (saved is not None and j == len(k.blocks) - 1) else self._buf(k.tag + f'.o{j}', M, dp, T))
            self._block_forward(blk, x, None, n_seq, M, bufs, train, seed, nxt)
            x = nxt
        L.gemm_nt(x, k.up.w, out, bias=k.up.b, R1=fus, M=M)

    def _kad_bufs(self, k, M):
        return dict(blk=[self._block_bufs(k.tag + f'.b{j}', blk, M, False) for j, blk in enumerate(k.blocks)],
                    bo=self._buf(k.tag + '.bo', M, k.dp, k.T), fus=self._buf(k.tag + '.fus', M, k.width, k.T))

    def _kad_backward(self, k, d_out, n_seq, M, train, seed, d_fus, saved):
        """d_fus = d_out + down^T(blocks^T(up^T d_out)); accumulates the weight gradients of down / up / the two blocks."""
        T, dp = k.T, k.dp
        self._dense_wgrad(k.up, d_out, saved['bo'], M)
        dx = self._buf(k.tag + '.dbo', M, dp, T)
        L.gemm_nt(d_out, k.up.wT, dx, M=M)
        for j in range(len(k.blocks) - 1, -1, -1):
            dprev = self._buf(k.tag + f'.dx{j}', M, dp, T)
            self._block_backward(k.blocks[j], dx, None, n_seq, M, saved['blk'][j], train, seed, dprev)
            dx = dprev
        self._dense_wgrad(k.down, dx, saved['fus'], M)
        L.gemm_nt(dx, k.down.wT, d_fus, R1=d_out, M=M)

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, rows, cols, dt):
        key = (name, cols, dt)
        t = self._bufs.get(key)
        if t is None or t.shape[0] < rows:
            t = torch.zeros(rows, cols, dtype=dt, device=self.dev)
            self._bufs[key] = t
        return t[:rows]

    def _buf_tail0(self, name, rows, cols, dt, real):
        """_buf for a gradient buffer whose producer writes only the first `real` rows while its consumers (dgrad GEMMs, ln_bwd
        column sums, gemm_tn weight gradients) run over all `rows` padded rows: rows [real, rows) must be EXACT zeros or a smaller
        batch that follows a larger one (run.py's DataLoader has no drop_last) sums the previous batch's rows into the adapter
        gradients.  A fresh buffer is zero; afterwards only what an earlier, larger call wrote has to be cleared, so the steady
        state (same batch size every step) costs no launch."""
        t = self._buf(name, rows, cols, dt)
        key = (name, cols, dt)
        hw = self._dirty.get(key, 0)
        if hw > real:
            full = self._bufs[key]
            full[real:min(hw, full.shape[0])].zero_()
        self._dirty[key] = real
        return t

    def _q8(self, blk):
        """This block's saved FFN activation derivative is the uint8 form (GELU FFN, bf16 storage)."""
        if blk.d_i is not None and (blk.d_i.trainable or blk.d_o2.trainable):
            return False             # --fine_tune_to all: dW_i / dx1 of a TRAINABLE FFN use the derivative at storage precision, not the 8-bit form
        return self.q8_deriv and blk.T == torch.bfloat16 and getattr(blk, 'ffn_act', L.ACT_GELU) == L.ACT_GELU and blk.F % 16 == 0

    Q8_TILED = _os.environ.get('A4R_Q8_TILED', '1') != '0'
    VSKIP = _os.environ.get('A4R_VSKIP', '1') != '0'

    def _vskip(self, blk, which):
        """Sub-layer `which` of a post-LN block keeps y = LN(v) per layer (the tensor the next GEMM reads anyway) instead of v, and the fused
        adapter backward rebuilds xhat = (y - beta) / gamma: 62 MB less written per fused forward launch at B = 32.  Needs the one-launch
        kernels (bf16, serial Houlsby / Compacter placement), a frozen LayerNorm and |gamma| bounded away from 0."""
        ad, pl, ln = (blk.ad1, blk.pl1, blk.ln1) if which == '1' else (blk.ad2, blk.pl2, blk.ln2)
        if not self.VSKIP or ad is None or pl != 'serial' or hasattr(blk, 'lnA') or blk.T != torch.bfloat16:
            return False
        if not (self.fuse_adapters and getattr(blk, 'Hv', blk.H) == blk.H and blk.H in (128, 256, 512, 768) and ad.dp == 64):
            return False
        if ln.g_gamma is not None or ln.g_beta is not None:
            return False
        key = '_vskip_ok' + which
        if not hasattr(blk, key):
            # xhat = (y - beta) / gamma from the bf16-rounded y carries an error of ~2^-9 |xhat + beta / gamma|: a column with a small
            # gamma next to a sizeable beta would put O(0.1 - 1) into xhat -- such layers keep v
            g, b = ln.gamma.detach().float(), ln.beta.detach().float()
            setattr(blk, key, bool(float(g.abs().min()) > 1e-3 and float((b / g).abs().max()) < 8.0))
        return getattr(blk, key)

    def _q8t(self, blk, M):
        """The 8-bit derivative tensor in the 256-tile kernel's own order (a4r_gemm_t.q8_tiled): only its writer (FFN-up) and its reader
        (the `* derivative` dgrad) ever touch it, and both see the same [M, F]."""
        return self.Q8_TILED and self._q8(blk) and M % 256 == 0 and blk.F % 256 == 0

    def _block_bufs(self, tag, blk, M, shared, Mc=None):
        """Activation buffers of one block: `shared` => transient set reused by every block (inference).
        Mc: row count of everything AFTER attention when only the CLS rows are carried on (last encoder layer)."""
        pre = tag if not shared else tag.split('.')[0] + '.shared'
        T, H, F = blk.T, blk.H, blk.F
        d = {}
        d['qkv'] = self._buf(pre + '.qkv', M, 3 * H, T)
        if getattr(blk, 'long', False):              # a4r_attn_long_*: row log-sum-exp, and the forward output for delta = dO . O
            d['lse'] = self._buf(pre + '.lse', (M // blk.S + 1) * blk.nh * blk.S, 1, torch.float32)
            if not shared:
                d['ctx_o'] = self._buf(pre + '.ctx_o', M, H, T)
        if Mc is not None:
            pre, M = pre + '.cls', Mc
        if (blk.lora or blk.pl1 == 'parallel' or blk.train_dense) and not shared:
            d['xin'] = self._buf(pre + '.xin', d['qkv'].shape[0], H, T)
        if (blk.pl2 == 'parallel' or blk.train_dense) and not shared:
            d['x1s'] = self._buf(pre + '.x1s', M, H, T)
        if blk.train_dense and not shared:           # inputs of attention.output.dense and output.dense (weight gradients)
            d['ctx_s'] = self._buf(pre + '.ctx_s', M, H, T)
            d['u_s'] = self._buf(pre + '.u_s', M, F, T)
        keep_y = not shared and Mc is None        # (training buffers of a full-row layer: the CLS-only last layer keeps v)
        d['h1'] = self._buf(pre + '.h1', M, H, T)
        d['y1' if keep_y and self._vskip(blk, '1') else 'v1'] = self._buf(pre + '.v1', M, H, T)
        d['st1'] = self._buf(pre + '.st1', M, 2, torch.float32)
        d['upre'] = self._buf(pre + '.upre', M, F, torch.uint8 if self._q8(blk) else T)
        d['h2'] = self._buf(pre + '.h2', M, H, T)
        d['y2' if keep_y and self._vskip(blk, '2') else 'v2'] = self._buf(pre + '.v2', M, H, T)
        d['st2'] = self._buf(pre + '.st2', M, 2, torch.float32)
        for k, ad, pl in (('1', blk.ad1, blk.pl1), ('2', blk.ad2, blk.pl2)):
            if ad is not None:
                d['zp' + k] = self._buf(pre + '.zp' + k, M, ad.dp, T)
                d['z' + k] = self._buf(pre + '.z' + k, M, ad.dp, T)
                if pl == 'pfeiffer':
                    d['t' + k] = self._buf(pre + '.t' + k, M, H, T)
                    d['va' + k] = self._buf(pre + '.va' + k, M, H, T)
                    d['sta' + k] = self._buf(pre + '.sta' + k, M, 2, torch.float32)
        return d

    @staticmethod
    def _vc(blk, t):
        """The valid columns of a zero-padded block's activation (LayerNorm width), the tensor itself otherwise."""
        hv = getattr(blk, 'Hv', blk.H)
        return t if hv == t.shape[1] else t[:, :hv]

    # ------------------------------------------------------------------ one block, forward
    def _sub_forward(self, blk, which, dense_in, w, bias, resid, ln, ad, pl, lnn, bufs, M, p_drop, site, seed, out, scales=None, out8=None):
        """dense -> dropout -> [adapter] -> LN(residual + .)  for the attention-output (which='1') or FFN-output ('2') half.
        scales = (scale_a, scale_b): dense_in / w are e4m3 operands (fp8 encoder).  out8 = (q, scale): `out` is ALSO wanted as e4m3 rows +
        per-row scales (the next fp8 GEMM's A operand): written by the fused adapter kernel, by one more row pass everywhere else."""
        h, v, st = bufs['h' + which], bufs.get('v' + which), bufs['st' + which]        # v None: this sub-layer keeps y (`out` IS bufs['y' + which])
        self._twin.pop(out.data_ptr(), None)           # (whatever fp32 twin an earlier sub-layer left for this buffer is stale from here on)
        assert v is not None or out.data_ptr() == bufs['y' + which].data_ptr()
        sk = dict(scale_a=scales[0], scale_b=scales[1]) if scales is not None else {}
        y8, ys = out8 if out8 is not None else (None, None)
        if not (ad is not None and pl not in ('pfeiffer', 'parallel') and self._fuse(blk, ad, h)):
            if out8 is not None:                   # (not the one-launch kernel: quantise `out` afterwards)
                self._sub_forward(blk, which, dense_in, w, bias, resid, ln, ad, pl, lnn, bufs, M, p_drop, site, seed, out, scales=scales)
                L.quant_rows_fp8(out, y8, ys, M=M)
                return
        # --residual_dtype fp32 on the sub-layers WITHOUT a one-launch serial adapter (un-adapted: Pfeiffer's attention half, LoRA, frozen,
        # soft prompt; Pfeiffer's FFN half): the dense output leaves the GEMM alone (bf16, as the reference's autocast Linear does), the
        # residual -- the fp32 twin the sub-layer below left, or the bf16 tensor itself at the first layer -- is added in fp32 inside
        # a4r_ln_fwd_sum, which normalises the unrounded sum and leaves the fp32 twin of its own output.
        r32mode = self.res32 and blk.T == torch.bfloat16 and getattr(blk, 'Hv', blk.H) == blk.H and v is not None and v.shape[1] == blk.H
        if ad is None:
            if r32mode:
                hd = self._buf('res32.h', M, blk.H, blk.T)
                L.gemm_nt(dense_in, w, hd, bias=bias, drop_p=p_drop, drop_site=site, drop_seed=seed, M=M, **sk)
                y32 = self._buf('res32.' + which, M, blk.H, torch.float32)
                L.ln_fwd_sum(hd, resid, ln.gamma, ln.beta, ln.eps, out, st, M=M, res32=self._twin_of(resid, M), sum_out=v, y32=y32)
                self._twin[out.data_ptr()] = y32
                return
            L.gemm_nt(dense_in, w, v, bias=bias, R1=resid, drop_p=p_drop, drop_site=site, drop_seed=seed, drop_first=True, M=M, **sk)
            L.ln_fwd(self._vc(blk, v), ln.gamma, ln.beta, ln.eps, self._vc(blk, out), st, M=M)
            return
        zp, z = bufs['zp' + which], bufs['z' + which]
        if pl == 'pfeiffer':          # model.py:321-329 / :458-471
            va, t, sta = bufs['va' + which], bufs['t' + which], bufs['sta' + which]
            if r32mode and self._fuse(blk, ad, t):
                hd = self._buf('res32.h', M, blk.H, blk.T)
                va32 = self._buf('res32.va', M, blk.H, torch.float32)
                L.gemm_nt(dense_in, w, hd, bias=bias, drop_p=p_drop, drop_site=site, drop_seed=seed, M=M, **sk)
                L.ln_fwd_sum(hd, resid, ln.gamma, ln.beta, ln.eps, t, sta, M=M, res32=self._twin_of(resid, M), sum_out=va, sum32=va32)   # va = h + input
                y32 = self._buf('res32.' + which, M, blk.H, torch.float32)
                L.adapter_ln_fwd(t, va, None, ad.wd, ad.bd, ad.wu, ad.bu, lnn.gamma, lnn.beta, lnn.eps, ad.act, zp, z, v, out, st, M=M, res32=va32, y32=y32, frag=ad.frag_f)
                self._twin[out.data_ptr()] = y32
                return
            L.gemm_nt(dense_in, w, va, bias=bias, R1=resid, drop_p=p_drop, drop_site=site, drop_seed=seed, drop_first=True, M=M, **sk)   # h + input
            L.ln_fwd(va, ln.gamma, ln.beta, ln.eps, t, sta, M=M)
            if self._fuse(blk, ad, t):
                L.adapter_ln_fwd(t, va, None, ad.wd, ad.bd, ad.wu, ad.bu, lnn.gamma, lnn.beta, lnn.eps, ad.act, zp, z, v, out, st, M=M, frag=ad.frag_f)
                return
            L.gemm_nt(t, ad.wd, z, bias=ad.bd, C2=zp, act=ad.act, M=M)
            L.gemm_nt(z, ad.wu, v, bias=ad.bu, R1=va, M=M)                   # adapter(t) + h + input
            L.ln_fwd(v, lnn.gamma, lnn.beta, lnn.eps, out, st, M=M)
            return
        if pl == 'parallel':          # model.py:265-270 / :474-520: LN(adapter(input) + dense_out + input), adapter has its inner residual
            L.gemm_nt(dense_in, w, h, bias=bias, R1=resid, drop_p=p_drop, drop_site=site, drop_seed=seed, drop_first=True, M=M, **sk)   # h + input
            L.gemm_nt(resid, ad.wd, z, bias=ad.bd, C2=zp, act=ad.act, M=M)
            L.gemm_nt(z, ad.wu, v, bias=ad.bu, R1=h, R2=resid, M=M)          # up + (h + input) + input
            L.ln_fwd(v, ln.gamma, ln.beta, ln.eps, out, st, M=M)
            return
        L.gemm_nt(dense_in, w, h, bias=bias, drop_p=p_drop, drop_site=site, drop_seed=seed, M=M, **sk)
        comp = ad.kind == 'compacter'  # no inner residual (modules.py:248-252); Houlsby: fc_up(act(fc_down(h))) + h, then + input (model.py:292-297)
        if self._fuse(blk, ad, h):     # ONE launch: down-projection, activation, up-projection, residual(s), LayerNorm (a4r_adapter_fused.hip)
            r32 = y32 = None
            if (self.res32 or self.res24) and blk.T == torch.bfloat16:
                # the residual stream in fp32 (reference under autocast: LayerNorm outputs fp32, BertSelfOutput's add promotes to it): this
                # sub-layer reads the fp32 twin of its residual input when the sub-layer below left one, and leaves the twin of its output
                # in one of two transient buffers (attention half / FFN half: a half's twin is dead once the next half of its kind has run).
                # bf24: the twin is a byte plane (int8) -- the next 8 mantissa bits beside the bf16 tensor.
                r32 = self._twin_of(resid, M)
                y32 = self._buf(('res8.' if self.res24 else 'res32.') + which, M, blk.H // self.lo_div if self.res24 else blk.H, torch.int8 if self.res24 else torch.float32)
            L.adapter_ln_fwd(h, resid if comp else h, None if comp else resid, ad.wd, ad.bd, ad.wu, ad.bu, ln.gamma, ln.beta, ln.eps, ad.act,
                             zp, z, v, out, st, M=M, y8=y8, ys=ys, **(dict(res32=r32, y32=y32) if y32 is not None else {}), frag=ad.frag_f)
            if y32 is not None:
                self._twin[out.data_ptr()] = y32
            return
        L.gemm_nt(h, ad.wd, z, bias=ad.bd, C2=zp, act=ad.act, M=M)
        if comp:
            L.gemm_nt(z, ad.wu, v, bias=ad.bu, R1=resid, M=M)
        else:
            L.gemm_nt(z, ad.wu, v, bias=ad.bu, R1=h, R2=resid, M=M)
        L.ln_fwd(v, ln.gamma, ln.beta, ln.eps, out, st, M=M)

    # ---- packed titles (self._pk, set per step by train_forward): item i owns token rows [off[i], off[i + 1]) of every [M, .] tensor of the item
    # tower -- its own token count instead of the batch's longest title (a4r_attn_t.offsets).  None: n_items x S rows as before.
    _pk = None

    def _off(self, blk):
        return self._pk['off'] if (self._pk is not None and getattr(blk, 'is_item', False)) else None

    def _cls_gather(self, src, dst, n_items, S, blk=None):
        """dst[i] = row of token 0 of item i"""
        if self._pk is not None and (blk is None or getattr(blk, 'is_item', False)):
            L.rows_idx_copy(src, dst, self._pk['off'], n_items)
        else:
            L.gather_rows(src, dst, n_items, S)

    def _cls_scatter_fill(self, src, dst, n_items, S, M, blk=None):
        """dst[row of token 0 of item i] = src[i], every other row < M zero"""
        if self._pk is not None and (blk is None or getattr(blk, 'is_item', False)):
            L.zero(dst[:M])
            L.rows_idx_copy(src, dst, self._pk['off'], n_items, scatter=True)
        else:
            L.scatter_rows_fill(src, dst, n_items, S, M)

    def _twin_of(self, t, M):
        """The fp32 twin of residual-stream tensor t ([>= M, H]) when the producing sub-layer left one in this forward (else None: the bf16
        tensor itself is the residual, e.g. the embedding output or a sub-layer that ran on the multi-launch path)."""
        w = self._twin.get(t.data_ptr())
        return w if (w is not None and w.shape[0] >= M and w.shape[1] == (t.shape[1] // self.lo_div if w.dtype == torch.int8 else t.shape[1])) else None

    def _fuse_bwd(self, blk, ad, t):
        return self._fuse(blk, ad, t) and blk.H != 1024          # (the backward kernel's LDS image of Wd does not fit at H = 1024)

    def _fuse(self, blk, ad, t):
        """The one-launch adapter kernels apply (bf16, bottleneck 64, a width they are instantiated for, no zero-padded block)."""
        return self.fuse_adapters and getattr(blk, 'Hv', blk.H) == blk.H and L.adapter_ln_ok(t, ad.dp)

    # ---- one C call per layer (a4r_encoder_layer_fwd / _bwd, ABI 409): the launches below, sequenced by the library for the layers it covers --
    # bf16, short attention, serial Houlsby adapters on the one-launch kernels on both halves, frozen backbone.  Bit-identical to the per-launch path
    # (tests/test_layer_call_gpu.py); A4R_LAYER_CALL=0 keeps the per-launch path everywhere.
    LAYER_CALL = _os.environ.get('A4R_LAYER_CALL', '1') != '0'
    _LAYER_DBG = _os.environ.get('A4R_LAYER_CALL', '1')          # (2: forward only, 3: backward only -- bisecting)

    def _layer_ok(self, blk, bufs, cls_rows, backward):
        if not (self.LAYER_CALL and hasattr(L, 'encoder_layer_fwd') and blk.T == torch.bfloat16 and not self.fp8 and not self.res32):
            return False
        if (self._LAYER_DBG == '2' and backward) or (self._LAYER_DBG == '3' and not backward):
            return False
        if cls_rows is not None or getattr(blk, 'long', False) or blk.lora or blk.train_dense or hasattr(blk, 'lnA') or getattr(blk, 'Hv', blk.H) != blk.H:
            return False
        if any(k in bufs for k in ('xin', 'x1s', 'ctx_s', 'u_s')) or blk.ffn_act != L.ACT_GELU or blk.S > 32 or blk.H not in (128, 256, 512, 768):
            return False
        if any(d is not None and d.trainable for d in blk.qkv):
            return False
        for ad, pl, ln in ((blk.ad1, blk.pl1, blk.ln1), (blk.ad2, blk.pl2, blk.ln2)):
            if ad is None or pl != 'serial' or ad.kind == 'compacter' or ad.virtual is not None or ad.dp != 64 or not self.fuse_adapters:
                return False
            if ln.g_gamma is not None or ln.g_beta is not None:
                return False
            if backward and ad.g_wu is not None and not (TN2 and TN2_BIAS and not WGRAD_STREAM and ad.s_wu is None and ad.s_bd is None and ad.g_bu is not None
                                                         and ad.g_bd is not None and self._bd_target(ad) is not None):
                return False
        return True

    def _layer_desc(self, blk, bufs, M):
        """The a4r_encoder_layer_t of (block, buffer set): static fields once, the per-step ones (seed, item count, masks, twins) by the caller."""
        key = (id(blk), id(bufs), M)
        cache = self.__dict__.setdefault('_layer_descs', {})
        hit = cache.get(key)
        if hit is not None and hit['bufs'] is bufs:        # (the entry keeps the buffer set alive: its id cannot be re-used by another one)
            return hit
        if len(cache) >= 256:                              # (a new buffer set per call under ragged batches / inference: bounded)
            cache.clear()
        T, H, F = blk.T, blk.H, blk.F
        d = L.EncoderLayer()
        d.M, d.H, d.F, d.S, d.n_heads, d.dh, d.causal = M, H, F, blk.S, blk.nh, blk.dh, int(blk.causal)
        d.scale, d.mask_neg, d.ln_eps = blk.scale, blk.mask_neg, blk.ln1.eps
        p = lambda t: None if t is None else t.data_ptr()
        d.wqkv, d.wqkvT, d.wo, d.woT, d.wi, d.wiT, d.wo2, d.wo2T = (p(t) for t in (blk.wqkv, blk.wqkvT, blk.wo, blk.woT, blk.wi, blk.wiT, blk.wo2, blk.wo2T))
        d.bqkv, d.bo, d.bi, d.bo2 = p(blk.bqkv), p(blk.bo), p(blk.bi), p(blk.bo2)
        d.ln1_g, d.ln1_b, d.ln2_g, d.ln2_b = p(blk.ln1.gamma), p(blk.ln1.beta), p(blk.ln2.gamma), p(blk.ln2.beta)
        keep = []
        for i, ad in enumerate((blk.ad1, blk.ad2)):
            a = d.ad[i]
            a.wd, a.wu, a.wdT, a.wuT, a.bd, a.bu, a.act = p(ad.wd), p(ad.wu), p(ad.wdT), p(ad.wuT), p(ad.bd), p(ad.bu), ad.act
            if ad.frag_f is not None:
                a.wd_f, a.wu_f, a.wuT_f, a.wdT_f = p(ad.frag_f[0]), p(ad.frag_f[1]), p(ad.frag_b[0]), p(ad.frag_b[1])
            # (the gradient targets are resolved per backward call: the flat gradient buffer they view is re-bound when an optimizer attaches)
        x1 = bufs['y1'] if 'y1' in bufs else self._buf('x1', M, H, T)
        ws = dict(ctx=self._buf('ctx', M, H, T), u=self._buf('u', M, F, T), x1=x1)
        d.qkv, d.ctx, d.h1, d.v1, d.zp1, d.z1 = p(bufs['qkv']), p(ws['ctx']), p(bufs['h1']), p(bufs.get('v1')), p(bufs['zp1']), p(bufs['z1'])
        d.u, d.upre, d.h2, d.v2, d.zp2, d.z2 = p(ws['u']), p(bufs['upre']), p(bufs['h2']), p(bufs.get('v2')), p(bufs['zp2']), p(bufs['z2'])
        d.st1, d.st2 = p(bufs['st1']), p(bufs['st2'])
        d.upre_q8, d.q8_tiled = int(self._q8(blk)), int(self._q8t(blk, M))
        hit = dict(d=d, x1=x1, keep=keep, ws=ws, bufs=bufs)
        cache[key] = hit
        return hit

    def _layer_forward(self, blk, x, key_mask, n_items, M, bufs, train, seed, x_out):
        e = self._layer_desc(blk, bufs, M)
        d, x1 = e['d'], e['x1']
        d.n_items, d.p_attn, d.p_hidden = n_items, (blk.p_attn if train else 0.0), (blk.p_hidden if train else 0.0)
        d.drop_site, d.drop_seed = blk.site, seed
        d.key_mask = None if key_mask is None else key_mask.data_ptr()
        off = self._off(blk)
        d.offsets = None if off is None else off.data_ptr()
        x_lo = self._twin_of(x, M) if self.res24 else None
        x1_lo = self._buf('res8.1', M, blk.H // self.lo_div, torch.int8) if self.res24 else None
        xo_lo = self._buf('res8.2', M, blk.H // self.lo_div, torch.int8) if self.res24 else None
        d.x_lo, d.x1_lo, d.xout_lo = (None if t is None else t.data_ptr() for t in (x_lo, x1_lo, xo_lo))
        d.lo_nibble = 1 if (self.res24 and self.lo_div == 2) else 0
        self._twin.pop(x1.data_ptr(), None)
        self._twin.pop(x_out.data_ptr(), None)
        if 'y2' in bufs:
            assert x_out.data_ptr() == bufs['y2'].data_ptr()
        L.encoder_layer_fwd(d, x, x1, x_out)
        if self.res24:
            self._twin[x1.data_ptr()], self._twin[x_out.data_ptr()] = x1_lo, xo_lo

    def _layer_backward(self, blk, dx_out, key_mask, n_items, M, bufs, train, seed, dx_in):
        e = self._layer_desc(blk, bufs, M)
        d, T, H, F = e['d'], blk.T, blk.H, blk.F
        d.n_items, d.p_attn, d.p_hidden = n_items, (blk.p_attn if train else 0.0), (blk.p_hidden if train else 0.0)
        d.drop_site, d.drop_seed = blk.site, seed
        d.key_mask = None if key_mask is None else key_mask.data_ptr()
        off = self._off(blk)
        d.offsets = None if off is None else off.data_ptr()
        self._wgrad_join()
        for i, ad in enumerate((blk.ad1, blk.ad2)):
            a = d.ad[i]
            if ad.g_wu is not None:
                gw, gd, gbu, gbd = ad.g_wu(), ad.g_wd(), ad.g_bu(), ad.g_bd()
                a.g_wu, a.g_wd, a.g_bu, a.g_bd, a.ldg_wu, a.ldg_wd = gw.data_ptr(), gd.data_ptr(), gbu.data_ptr(), gbd.data_ptr(), gw.stride(0), gd.stride(0)
            else:
                a.g_wu = a.g_wd = a.g_bu = a.g_bd = None
        n_tok = self._pk['Mtok'] if off is not None else n_items * blk.S
        sc = dict(dv1=self._buf('dv1', M, H, T), dv2=self._buf('dv2', M, H, T), dzp=self._buf('dzp', M, 64, T), d_h=self._buf('dh2', M, H, T),
                  du=self._buf('du', M, F, T), dx1=self._buf('dx1', M, H, T), dctx=self._buf('dctx', M, H, T))
        for k, t in sc.items():
            setattr(d, k, t.data_ptr())
        d.dqkv = self._buf_tail0('dqkv', M, 3 * H, T, n_tok).data_ptr() if dx_in is not None else self._buf('dqkv', M, 3 * H, T).data_ptr()
        x_out = bufs['y2'] if 'y2' in bufs else bufs['v2']           # (with v2 kept the library never reads x_out)
        L.encoder_layer_bwd(d, e['x1'], x_out, dx_out, dx_in)

    def _block_forward(self, blk, x, key_mask, n_items, M, bufs, train, seed, x_out, cls_rows=None, x8=None, want8=False):
        """cls_rows = Ip: after attention only row 0 of every item (the CLS token, all the item head reads,
        model/encoders.py:55) is carried through attention-output, FFN and adapters: x_out is then [Ip, H].
        fp8 encoder (post-LN text tower): x8 = (e4m3 rows, row scales) of x when the layer below produced them; want8: return the same
        for x_out (None otherwise).  Frozen qkv / attention-output / FFN-up / FFN-down run on e4m3 operands, everything else as in bf16."""
        T, H = blk.T, blk.H
        if x8 is None and not want8 and self._layer_ok(blk, bufs, cls_rows, False):
            self._layer_forward(blk, x, key_mask, n_items, M, bufs, train, seed, x_out)
            return None
        pa = blk.p_attn if train else 0.0
        ph = blk.p_hidden if train else 0.0
        f8 = self.fp8 and T == torch.bfloat16 and M % 256 == 0 and getattr(blk, 'wqkv8', None) is not None
        if f8:
            if x8 is None:
                x8 = (self._buf('x8', M, H, torch.uint8), self._buf('x8s', M, 1, torch.float32))
                L.quant_rows_fp8(x, x8[0], x8[1], M=M)
            L.gemm_nt(x8[0], blk.wqkv8, bufs['qkv'], bias=blk.bqkv, M=M, scale_a=x8[1], scale_b=blk.wqkv8s)
        else:
            L.gemm_nt(x, blk.wqkv, bufs['qkv'], bias=blk.bqkv, M=M)
        if 'xin' in bufs and x.data_ptr() != bufs['xin'].data_ptr():
            L.gather_rows(x, bufs['xin'], M, 1)      # LoRA backward needs the block input (t = x A^T, dA = dt^T x)
        if getattr(blk, 'long', False):
            ctx = bufs['ctx_o'] if 'ctx_o' in bufs else self._buf('ctx', M, H, T)
            L.attn_long_fwd(bufs['qkv'], ctx, bufs['lse'], n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.scale,
                            drop_p=pa, drop_site=blk.site, drop_seed=seed, key_mask=key_mask, causal=blk.causal)
        else:
            # (trainable attention output: its weight gradient needs ctx per layer -- the attention kernel writes the kept buffer directly)
            ctx = bufs['ctx_s'] if ('ctx_s' in bufs and cls_rows is None) else self._buf('ctx', M, H, T)
            L.attn_fwd(bufs['qkv'], ctx, key_mask, n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.causal, blk.scale, blk.mask_neg,
                       drop_p=pa, drop_site=blk.site, drop_seed=seed, offsets=self._off(blk))
        if cls_rows is not None:
            ctx_c, x_c = self._buf('ctx_c', cls_rows, H, T), self._buf('x_c', cls_rows, H, T)
            self._cls_gather(ctx, ctx_c, n_items, blk.S, blk)
            self._cls_gather(x, x_c, n_items, blk.S, blk)
            x32 = self._twin_of(x, M) if (self.res32 or self.res24) else None
            if x32 is not None:                        # the CLS rows of the fp32 residual stream too
                if x32.dtype == torch.int8:            # (byte plane: its rows move as H / 2 two-byte elements)
                    x_c32 = self._buf('x_c8', cls_rows, x32.shape[1], torch.int8)
                    self._cls_gather(x32.view(torch.bfloat16), x_c32.view(torch.bfloat16), n_items, blk.S, blk)
                else:
                    x_c32 = self._buf('x_c32', cls_rows, H, torch.float32)
                    self._cls_gather(x32, x_c32, n_items, blk.S, blk)
                self._twin[x_c.data_ptr()] = x_c32
            ctx, x, M = ctx_c, x_c, cls_rows
        if 'ctx_s' in bufs and ctx is not bufs['ctx_s']:
            L.gather_rows(ctx, bufs['ctx_s'], M, 1)
        x1 = self._buf('x1', M, H, T)
        x1 = bufs['x1s'] if 'x1s' in bufs else x1
        if 'y1' in bufs:                           # kept per layer: backward rebuilds xhat from it (_vskip)
            if 'x1s' in bufs:
                raise RuntimeError('y1 and x1s are the same tensor: one of them should not have been allocated')
            x1 = bufs['y1']
        f8 = self.fp8 and T == torch.bfloat16 and M % 256 == 0             # (M: the CLS rows in the last layer)
        f8_o = f8 and getattr(blk, 'wo8', None) is not None and 'ctx_s' not in bufs
        f8_up = f8 and getattr(blk, 'wi8', None) is not None and blk.ffn_act == L.ACT_GELU
        f8_ffn = f8_up and getattr(blk, 'wo28', None) is not None and 'u_s' not in bufs and self._q8(blk)
        x1_8 = (self._buf('x1_8', M, H, torch.uint8), self._buf('x1_8s', M, 1, torch.float32)) if f8_up else None
        if f8_o:                                   # the attention output as e4m3 + per-token scale (one row pass), then the fp8 GEMM
            c8, cs = self._buf('ctx8', M, H, torch.uint8), self._buf('ctx8s', M, 1, torch.float32)
            L.quant_rows_fp8(ctx, c8, cs, M=M)
            self._sub_forward(blk, '1', c8, blk.wo8, blk.bo, x, blk.ln1, blk.ad1, blk.pl1, blk.lnn1, bufs, M, ph, blk.site + 1, seed, x1,
                              scales=(cs, blk.wo8s), out8=x1_8)
        else:
            self._sub_forward(blk, '1', ctx, blk.wo, blk.bo, x, blk.ln1, blk.ad1, blk.pl1, blk.lnn1, bufs, M, ph, blk.site + 1, seed, x1, out8=x1_8)
        out8 = (self._buf('x8', M, H, torch.uint8), self._buf('x8s', M, 1, torch.float32)) if (want8 and f8) else None
        if f8_ffn:         # FFN-up writes u as e4m3 with a static scale (+ the 8-bit derivative), FFN-down reads it
            u = self._buf('u8', M, blk.F, torch.uint8)
            L.gemm_nt(x1_8[0], blk.wi8, u, bias=blk.bi, C2=bufs['upre'], act=L.ACT_GELU, c2_deriv='q8', M=M, scale_a=x1_8[1], scale_b=blk.wi8s,
                      c_fp8=1, c_scale=self.FP8_U_SCALE, q8_tiled=self._q8t(blk, M))
            self._sub_forward(blk, '2', u, blk.wo28, blk.bo2, x1, blk.ln2, blk.ad2, blk.pl2, blk.lnn2, bufs, M, ph, blk.site + 2, seed, x_out,
                              scales=(self._const_rows('su', M, self.FP8_U_SCALE), blk.wo28s), out8=out8)
            return out8
        u = bufs['u_s'] if 'u_s' in bufs else self._buf('u', M, blk.F, T)
        if f8_up:
            L.gemm_nt(x1_8[0], blk.wi8, u, bias=blk.bi, C2=bufs['upre'], act=L.ACT_GELU, c2_deriv='q8' if self._q8(blk) else True, M=M,
                      scale_a=x1_8[1], scale_b=blk.wi8s, q8_tiled=self._q8t(blk, M))
        else:
            L.gemm_nt(x1, blk.wi, u, bias=blk.bi, C2=bufs['upre'], act=blk.ffn_act, c2_deriv='q8' if self._q8(blk) else True, M=M,      # 'upre' holds act'(pre)
                      q8_tiled=self._q8t(blk, M))
        self._sub_forward(blk, '2', u, blk.wo2, blk.bo2, x1, blk.ln2, blk.ad2, blk.pl2, blk.lnn2, bufs, M, ph, blk.site + 2, seed, x_out, out8=out8)
        return out8

    # ------------------------------------------------------------------ one block, backward
    def _sub_backward(self, blk, which, dy, ln, ad, pl, lnn, bufs, M, p_drop, site, seed):
        """Backward of _sub_forward.  Returns (d_dense_out, d_resid): gradient wrt the dense output (already through
        its dropout mask) and wrt the residual input.  Writes adapter / LN parameter gradients."""
        T, H = blk.T, blk.H
        v, st = bufs.get('v' + which), bufs['st' + which]
        beta_y = None
        if v is None:                                    # the forward kept y = LN(v) instead (see _vskip)
            v, beta_y = bufs['y' + which], ln.beta
        gg = lambda f: f() if f is not None else None
        self._wgrad_join()                               # dzp / dv are about to be overwritten
        if ad is None:
            dv = self._buf('dv' + which, M, H, T)
            if p_drop > 0 and getattr(blk, 'Hv', H) == H:          # one launch writes dv and mask * dv (a4r_ln_bwd's second output; mask index = row * H + col)
                dh = self._buf('dh' + which, M, H, T)
                L.ln_bwd(dy, v, st, ln.gamma, dv, M=M, dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta),
                         dv2=dh, drop2_p=p_drop, drop2_site=site, drop2_seed=seed)
                return dh, dv
            L.ln_bwd(self._vc(blk, dy), self._vc(blk, v), st, ln.gamma, self._vc(blk, dv), M=M, dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta))
            if p_drop > 0:
                dh = self._buf('dh' + which, M, H, T)
                L.dropout_apply(dv, dh, p_drop, site, seed, M=M)
                return dh, dv
            return dv, dv
        zp, z = bufs['zp' + which], bufs['z' + which]
        dzp = self._buf('dzp', M, ad.dp, T)
        dv = self._buf('dv' + which, M, H, T)
        dh = self._buf('dh' + which, M, H, T)
        if pl == 'pfeiffer':
            va, t, sta = bufs['va' + which], bufs['t' + which], bufs['sta' + which]
            dt = self._buf('dt', M, H, T)
            fused_bd = False
            b2 = False
            if self._fuse_bwd(blk, ad, dy):
                b2 = self._tn2_bias_ok(ad, dv, M)
                L.adapter_ln_bwd(dy, v, st, lnn.gamma, None, zp, ad.act, ad.wuT, ad.wdT, False, dv, dzp, dt,
                                 dgamma=gg(lnn.g_gamma), dbeta=gg(lnn.g_beta), dbias=None if b2 else gg(ad.g_bu), M=M, dbd=None if b2 else self._bd_target(ad), frag=ad.frag_b)
                fused_bd = self._bd_target(ad) is not None
            else:
                L.ln_bwd(dy, v, st, lnn.gamma, dv, M=M, dgamma=gg(lnn.g_gamma), dbeta=gg(lnn.g_beta), dbias=gg(ad.g_bu))
                L.gemm_nt(dv, ad.wuT, dzp, Pre=zp, dact=ad.act, M=M)
                L.gemm_nt(dzp, ad.wdT, dt, M=M)
            self._adapter_wgrads(ad, dv, z, dzp, t, M, bd_done=fused_bd, bias_in_tn2=b2)
            dva = self._buf('dva', M, H, T)
            if p_drop > 0:
                L.ln_bwd(dt, va, sta, ln.gamma, dva, M=M, dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta), dres=dv,
                         dv2=dh, drop2_p=p_drop, drop2_site=site, drop2_seed=seed)
                return dh, dva
            L.ln_bwd(dt, va, sta, ln.gamma, dva, M=M, dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta), dres=dv)
            return dva, dva
        h = bufs['h' + which]
        if pl != 'parallel' and self._fuse_bwd(blk, ad, dy):
            # ONE launch: LayerNorm backward, dzp = (dv Wu) * act'(zp), dh = mask * (dzp Wd [+ dv]) (a4r_adapter_fused.hip)
            b2 = self._tn2_bias_ok(ad, dv, M)
            L.adapter_ln_bwd(dy, v, st, ln.gamma, None, zp, ad.act, ad.wuT, ad.wdT, ad.kind != 'compacter', dv, dzp, dh,
                             dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta), dbias=None if b2 else gg(ad.g_bu), M=M,
                             drop_p=p_drop, drop_site=site, drop_seed=seed, dbd=None if b2 else self._bd_target(ad), beta_y=beta_y, frag=ad.frag_b)
            self._adapter_wgrads(ad, dv, z, dzp, h, M, bd_done=self._bd_target(ad) is not None, bias_in_tn2=b2)
            return dh, dv
        assert beta_y is None, 'a sub-layer that keeps y instead of v runs the fused backward only'
        L.ln_bwd(dy, v, st, ln.gamma, dv, M=M, dgamma=gg(ln.g_gamma), dbeta=gg(ln.g_beta), dbias=gg(ad.g_bu))
        L.gemm_nt(dv, ad.wuT, dzp, Pre=zp, dact=ad.act, M=M)
        if pl == 'parallel':
            sub_in = bufs['xin'] if which == '1' else bufs['x1s']
            dres = self._buf('dres' + which, M, H, T)
            L.gemm_nt(dzp, ad.wdT, dres, R1=dv, R2=dv, M=M)            # adapter path + its inner residual + the outer residual
            self._adapter_wgrads(ad, dv, z, dzp, sub_in, M)
            if p_drop > 0:
                L.dropout_apply(dv, dh, p_drop, site, seed, M=M)
                return dh, dres
            return dv, dres
        if ad.kind == 'compacter':
            L.gemm_nt(dzp, ad.wdT, dh, drop_p=p_drop, drop_site=site, drop_seed=seed, M=M)
        else:
            L.gemm_nt(dzp, ad.wdT, dh, R1=dv, drop_p=p_drop, drop_site=site, drop_seed=seed, M=M)
        self._adapter_wgrads(ad, dv, z, dzp, h, M)
        return dh, dv

    LORA_FUSED = _os.environ.get('A4R_LORA_FUSED', '1') != '0'       # (0: the five separate products, A/B runs)

    def _lora_backward_all(self, blk, dqkv, x, M):
        """Low-rank gradients of every LoRA of a block.  Two small-rank LoRAs (the image tower's q, v) share the launches that read x:
        t = x [A_q ; A_v]^T once, dt = (dq B_q + dv B_v) s accumulated into one [M, 64] buffer, dA = dt^T x once, the two dB = d.^T t
        products in one a4r_gemm_tn2 launch, and the two bias gradients come out of that launch too (a column of ones in t) -- 5 launches and
        2 passes over x instead of 10 and 4."""
        sh = blk.lora[0].share if blk.lora else None
        if sh is None or any(lo.share is not sh for lo in blk.lora):
            for lo in blk.lora:
                self._lora_backward(blk, lo, dqkv, x, M)
            return
        H, T = blk.H, blk.T
        a, b = blk.lora
        dqa, dqb = dqkv[:, a.slot * H:(a.slot + 1) * H], dqkv[:, b.slot * H:(b.slot + 1) * H]
        if self.LORA_FUSED and a.r <= 15 and b.r <= 15 and L.lora_bwd_fused_ok(x, M, H):
            # round 4: all of the below in ONE pass over x, dq, dv (a4r_lora_bwd_fused: 306 MB instead of 612 MB per layer at the image tower's rows);
            # ranks <= 8 share one rank tile, 9 - 15 (CV/run_adapter.py's hard-coded 12) get one each
            oc, R = _Lora.ONES_COL, (8 if a.r <= 8 and b.r <= 8 else 16)
            L.lora_bwd_fused(x, dqa, dqb, sh['A'][a.off:a.off + R], sh['A'][b.off:b.off + R], a.BT[a.off:a.off + R], b.BT[b.off:b.off + R],
                             a.scaling, b.scaling, sh['s_A'][a.off:a.off + R], sh['s_A'][b.off:b.off + R],
                             a.s_B[:, a.off:a.off + R], b.s_B[:, b.off:b.off + R],
                             a.s_B[:, oc] if a.g_bias is not None else None, b.s_B[:, oc] if b.g_bias is not None else None, M, rank_rows=R)
            return
        t = self._buf('lora_t', M, 64, T)
        dt = self._buf('lora_dt', M, 64, T)
        L.gemm_nt(x, sh['A'], t, bias=sh['ones'], M=M)                 # t[:, off .. off + r] per LoRA; t[:, ONES_COL] = 1 (bias gradients, see _Lora)
        L.gemm_nt(dqa, a.BT, dt, alpha=a.scaling, M=M)                 # dt = (dq B_q) s        (columns 0 .. r)
        L.gemm_nt(dqb, b.BT, dt, alpha=b.scaling, R1=dt, M=M)          #    + (dv B_v) s        (columns 16 .. 16 + r)
        if TN2 and T == torch.bfloat16 and M % 64 == 0:
            L.gemm_tn2(dqa, t, a.s_B, dqb, t, b.s_B, M=M)              # dB_. = d.^T t (the corner of its own rank columns is flushed)
        else:
            L.gemm_tn(dqa, t, a.s_B, M=M)
            L.gemm_tn(dqb, t, b.s_B, M=M)
        L.gemm_tn(dt, x, sh['s_A'], M=M)                               # dA rows off .. off + r per LoRA

    def _lora_backward(self, blk, lo, dqkv, x, M):
        H, T = blk.H, blk.T
        dq = dqkv[:, lo.slot * H:(lo.slot + 1) * H]
        if lo.g_bias is not None:
            L.colsum(dq, lo.g_bias(), M=M)
        if lo.r == 0:
            if lo.g_W is not None:
                L.gemm_tn(dq, x, lo.g_W(), M=M)                  # dW = dq^T x
            return
        t = self._buf('lora_t', M, lo.rp, T)
        dt = self._buf('lora_dt', M, lo.rp, T)
        L.gemm_nt(x, lo.A, t, M=M)                               # t  = x A^T
        L.gemm_nt(dq, lo.BT, dt, alpha=lo.scaling, M=M)          # dt = (dq B) s
        L.gemm_tn(dq, t, lo.s_B, M=M)                            # dB = dq^T t s   (scratch: cleared at the start of backward, its valid
        L.gemm_tn(dt, x, lo.s_A, M=M)                            # dA = dt^T x      corner added to the flat gradient by _flush_corners)

    WGRAD_SIDE_OK = True            # the text tower's backward joins the side stream before dv / dzp are reused (_sub_backward)

    def _wgrad_join(self):
        """Main stream waits for the weight-gradient kernels that were put on the side stream (no-op without one)."""
        if self._wdone is not None:
            torch.cuda.current_stream().wait_event(self._wdone)
            self._wdone = None

    def _bd_target(self, ad):
        """Where the down-projection's bias gradient accumulates (64 floats: the flat gradient or its zero-padded scratch), or None."""
        if ad.g_bd is None or not FUSE_BD:
            return None
        return ad.s_bd if ad.s_bd is not None else ad.g_bd()

    def _tn2_bias_ok(self, ad, dv, M):
        """The adapter's two bias gradients can ride in its weight-gradient launch (a4r_gemm_tn2's xsum outputs: db_up = colsum(dv),
        db_down = colsum(dzp), from the bf16 tensors that launch reads anyway) instead of in the fused backward kernel's end-of-launch
        flush -- 832 atomics from each of its 256 workgroups onto the same addresses, 6 - 9 us per launch (A4R_TN2_BIAS=0: the flush)."""
        return (TN2_BIAS and TN2 and not (WGRAD_STREAM and self.WGRAD_SIDE_OK) and dv.dtype == torch.bfloat16 and M % 64 == 0
                and (ad.virtual is not None or ad.g_wu is not None) and ad.s_bd is None and ad.g_bu is not None and ad.g_bd is not None
                and self._bd_target(ad) is not None and ad.dp == 64 and dv.shape[1] % 64 == 0)

    def _adapter_wgrads(self, ad, dv, z, dzp, down_in, M, bd_done=False, bias_in_tn2=False):
        """dW_up = dv^T z, dW_down = dzp^T down_in, db_down = colsum(dzp)  (db_up comes from ln_bwd's dbias; bd_done: the fused
        backward kernel has already accumulated db_down; bias_in_tn2: neither was accumulated, both ride in this launch).
        A4R_WGRAD_STREAM=1: on a side stream, to run in the tail rounds of the dgrad GEMMs that follow (nothing on the
        dgrad chain reads these results); joined before dv / dzp are reused and at the end of the backward pass."""
        if ad.virtual is None and ad.g_wu is None:
            return                       # frozen adapter: nothing to accumulate
        if WGRAD_STREAM and self.WGRAD_SIDE_OK and ad.s_wu is None and ad.virtual is None and torch.device(self.dev).type == 'cuda':
            if self._wstream is None:
                self._wstream = torch.cuda.Stream(device=self.dev)
                self._wev = [torch.cuda.Event(), torch.cuda.Event()]
            ev = self._wev[0]
            ev.record()
            with torch.cuda.stream(self._wstream):
                self._wstream.wait_event(ev)
                if TN2 and dv.dtype == torch.bfloat16:    # both products in one launch (32 against 2 x 21 us)
                    L.gemm_tn2(dv, z, ad.g_wu(), dzp, down_in, ad.g_wd(), M=M)
                else:
                    L.gemm_tn(dv, z, ad.g_wu(), M=M)
                    L.gemm_tn(dzp, down_in, ad.g_wd(), M=M)
                if ad.g_bd is not None and ad.s_bd is None and not bd_done:
                    L.colsum(dzp, ad.g_bd(), M=M)
                self._wev[1].record()
            self._wdone = self._wev[1]
            self._wev.reverse()
            if ad.g_bd is not None and ad.s_bd is not None and not bd_done:
                L.colsum(dzp, ad.s_bd, M=M)
            return
        # zero-padded (d < 64) or virtual (Compacter) matrices: into the scratch arena (cleared at the start of backward; the valid
        # corners reach the flat gradient through _flush_corners / a4r_phm_bwd at its end)
        t_wu, t_wd = (ad.s_wu if ad.s_wu is not None else ad.g_wu()), (ad.s_wd if ad.s_wd is not None else ad.g_wd())
        if bias_in_tn2:
            assert dv.shape[1] * z.shape[1] == dzp.shape[1] * down_in.shape[1]
            L.gemm_tn2(dv, z, t_wu, dzp, down_in, t_wd, M=M, xsum1=ad.g_bu(), xsum2=ad.g_bd())
            return
        if TN2 and dv.dtype == torch.bfloat16 and dv.shape[1] * z.shape[1] == dzp.shape[1] * down_in.shape[1] and M % 64 == 0:
            L.gemm_tn2(dv, z, t_wu, dzp, down_in, t_wd, M=M)
        else:
            L.gemm_tn(dv, z, t_wu, M=M)
            L.gemm_tn(dzp, down_in, t_wd, M=M)
        if ad.g_bd is not None and not bd_done:
            L.colsum(dzp, ad.s_bd if ad.s_bd is not None else ad.g_bd(), M=M)

    def _block_backward(self, blk, dx_out, key_mask, n_items, M, bufs, train, seed, dx_in, cls_rows=None):
        T, H, F = blk.T, blk.H, blk.F
        if self._layer_ok(blk, bufs, cls_rows, True):
            self._layer_backward(blk, dx_out, key_mask, n_items, M, bufs, train, seed, dx_in)
            return
        pa = blk.p_attn if train else 0.0
        ph = blk.p_hidden if train else 0.0
        M_full = M
        if cls_rows is not None:
            M = cls_rows
        dh2, dres2 = self._sub_backward(blk, '2', dx_out, blk.ln2, blk.ad2, blk.pl2, blk.lnn2, bufs, M, ph, blk.site + 2, seed)
        self._dense_wgrad(blk.d_o2, dh2, bufs.get('u_s'), M)
        dx1 = self._buf('dx1', M, H, T)
        if (self.fp8 and getattr(blk, 'wo2T8', None) is not None and T == torch.bfloat16 and M % 256 == 0 and bufs['upre'].dtype == torch.uint8
                and 'u_s' not in bufs):
            # frozen FFN, both dgrads on e4m3 operands with ONE scale per token row carried through the chain (_build_fp8):
            #   dh2 -> e4m3 + row scale | du = (dh2 W2) * gelu' leaves its GEMM as e4m3 with scale[m] * c_du | dx1 = du W1 + dres2 (bf16 out)
            do8, dos = self._buf('do8', M, H, torch.uint8), self._buf('do8s', M, 1, torch.float32)
            L.quant_rows_fp8(dh2, do8, dos, M=M)
            du8, dus = self._buf('du8', M, F, torch.uint8), self._buf('du8s', M, 1, torch.float32)
            L.gemm_nt(do8, blk.wo2T8, du8, Pre=bufs['upre'], dact=L.DACT_MUL_Q8, M=M, scale_a=dos, scale_b=blk.wo2T8s,
                      c_fp8=2, c_scale=blk.c_du, c_scale_out=dus, q8_tiled=self._q8t(blk, M))
            L.gemm_nt(du8, blk.wiT8, dx1, R1=dres2, M=M, scale_a=dus, scale_b=blk.wiT8s)
        else:
            du = self._buf('du', M, F, T)
            L.gemm_nt(dh2, blk.wo2T, du, Pre=bufs['upre'], dact=L.DACT_MUL_Q8 if self._q8(blk) else L.DACT_MUL, M=M, q8_tiled=self._q8t(blk, M))
            self._dense_wgrad(blk.d_i, du, bufs.get('x1s'), M)
            L.gemm_nt(du, blk.wiT, dx1, R1=dres2, M=M)
        dh1, dres1 = self._sub_backward(blk, '1', dx1, blk.ln1, blk.ad1, blk.pl1, blk.lnn1, bufs, M, ph, blk.site + 1, seed)
        qkv_train = any(d is not None and d.trainable for d in blk.qkv)
        pend = []                                                  # the attention output's weight gradient rides in the q / k / v launch below
        if qkv_train and cls_rows is None:                         # (same token rows; dh1 and ctx_s are not written in between)
            pend = [(blk.d_o, dh1, bufs.get('ctx_s'))]
        else:
            self._dense_wgrad(blk.d_o, dh1, bufs.get('ctx_s'), M)
        if dx_in is None and not blk.lora and not qkv_train:      # first encoder layer: nothing trainable sits below its attention
            return
        dctx = self._buf('dctx_c' if cls_rows is not None else 'dctx', M, H, T)
        L.gemm_nt(dh1, blk.woT, dctx, M=M)
        if cls_rows is not None:                   # back to token rows: gradients live on the CLS rows only
            M = M_full
            full = self._buf('dctx', M, H, T)
            self._cls_scatter_fill(dctx, full, n_items, blk.S, M, blk)  # CLS rows written, every other row zeroed, one pass
            dctx = full
            rfull = self._buf('dres_full', M, H, T)
            self._cls_scatter_fill(dres1, rfull, n_items, blk.S, M, blk)
            dres1 = rfull
        n_tok = self._pk['Mtok'] if self._off(blk) is not None else n_items * blk.S
        dqkv = self._buf_tail0('dqkv', M, 3 * H, T, n_tok)                 # attn_bwd writes the real token rows only
        if getattr(blk, 'long', False):
            ws = self._buf('attn_ws', bufs['lse'].shape[0], 1, torch.float32)
            L.attn_long_bwd(bufs['qkv'], bufs['ctx_o'], dctx, dqkv, bufs['lse'], ws, n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.scale,
                            drop_p=pa, drop_site=blk.site, drop_seed=seed, key_mask=key_mask, causal=blk.causal)
        else:
            L.attn_bwd(bufs['qkv'], dctx, dqkv, key_mask, n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.causal, blk.scale, blk.mask_neg,
                       drop_p=pa, drop_site=blk.site, drop_seed=seed, offsets=self._off(blk))
        if blk.lora:
            self._lora_backward_all(blk, dqkv, bufs['xin'], M)
        self._dense_wgrads([(d, dqkv[:, sl * H:(sl + 1) * H], bufs.get('xin')) for sl, d in enumerate(blk.qkv)] + pend, M)
        if dx_in is not None:
            L.gemm_nt(dqkv, blk.wqkvT, dx_in, R1=dres1, M=M)

    def _dense_wgrads(self, items, M):
        """_dense_wgrad for up to four (Linear, dy, x) over the same M token rows: one a4r_gemm_tn_multi call (large bf16 products: ONE launch)."""
        live = [(d, dy, x) for d, dy, x in items if d is not None and d.trainable]
        if len(live) < 2 or len(live) > 4 or any(d.g_w is None for d, _, _ in live):
            for d, dy, x in live:
                self._dense_wgrad(d, dy, x, M)
            return
        L.gemm_tn_multi([(dy, x, d.s_w if d.s_w is not None else d.g_w().view(d.out_f, d.in_f),
                          None if d.g_b is None else (d.s_b if d.s_b is not None else d.g_b())) for d, dy, x in live], M=M)

    def _dense_wgrad(self, d, dy, x, M):
        """dW += dy^T x, db += column sums of dy for a trainable backbone Linear (--fine_tune_to all)."""
        if d is None or not d.trainable:
            return
        if d.g_w is not None and d.g_b is not None:               # one pass over dy for both (large bf16 outputs: one launch, a4r_gemm_tn256.hip)
            L.gemm_tn_bias(dy, x, d.s_w if d.s_w is not None else d.g_w().view(d.out_f, d.in_f),
                           d.s_b if d.s_b is not None else d.g_b(), M=M)
            return
        if d.g_w is not None:
            L.gemm_tn(dy, x, d.s_w if d.s_w is not None else d.g_w().view(d.out_f, d.in_f), M=M)       # (zero-padded storage: scratch + corner)
        if d.g_b is not None:
            L.colsum(dy, d.s_b if d.s_b is not None else d.g_b(), M=M)

    # ------------------------------------------------------------------ K-Adapter chains
    def _kad_chain_forward(self, x_last, n_items, M, Ip, train, seed, cls, keep):
        """model.py:546-559: last = 0; for each listed hidden state: last = adapter(hidden + last); cls rows of
        com_dense([last_hidden ; last]) -> `cls`."""
        H, T, nb = self.H, self.T, len(self.bert_blocks)
        if keep and (getattr(self, '_kad_saved_b', None) is None or self._kad_saved_M != M):
            self._kad_saved_b = [self._kad_bufs(k, M) for k in self.bert_kads]
            self._kad_saved_M = M
        last = None
        for j, (kk, kad) in enumerate(zip(self.bert_klist, self.bert_kads)):
            sv = self._kad_saved_b[j] if keep else None
            fus = sv['fus'] if sv is not None else self._buf(kad.tag + '.fus', M, H, T)
            fus.copy_(x_last if kk == nb else self._buf(f'khs{kk}', M, H, T))
            if last is not None:
                fus.add_(last)
            out = self._buf(kad.tag + '.out', M, H, T)
            self._kad_forward(kad, fus, n_items, M, train, seed, out, sv)
            last = out
        cat = self._buf('kcat', Ip, 2 * H, T)
        L.gather_rows(x_last, cat[:, :H], n_items, self.S)
        L.gather_rows(last, cat[:, H:], n_items, self.S)
        L.gemm_nt(cat, self.d_com.w, cls, bias=self.d_com.b, M=Ip)

    def _kad_chain_backward(self, dcls, n_items, M, Ip, train, seed, dxb):
        """-> {hidden-state index: gradient buffer} for the backbone (only used when something inside it trains); dxb receives
        the gradient of the last hidden state that came through com_dense."""
        H, T, S = self.H, self.T, self.S
        self._dense_wgrad(self.d_com, dcls, self._buf('kcat', Ip, 2 * H, T), Ip)
        dcat = self._buf('dkcat', Ip, 2 * H, T)
        L.gemm_nt(dcls, self.d_com.wT, dcat, M=Ip)
        dxb.zero_()
        L.scatter_rows(dcat[:, :H], dxb, n_items, S)
        dlast = self._buf('kd_last', M, H, T)
        dlast.zero_()
        L.scatter_rows(dcat[:, H:], dlast, n_items, S)
        d_hs = {}
        for j in range(len(self.bert_kads) - 1, -1, -1):
            kad, kk = self.bert_kads[j], self.bert_klist[j]
            d_fus = self._buf(kad.tag + '.dfus', M, H, T)
            self._kad_backward(kad, dlast, n_items, M, train, seed, d_fus, self._kad_saved_b[j])
            d_hs[kk] = d_fus if kk not in d_hs else d_hs[kk].add_(d_fus)
            dlast = d_fus
        return d_hs

    # ------------------------------------------------------------------ item tower / user tower
    def _encode(self, news, n_items, train, seed, saved):
        """news [n, 2S] int64 (ids || mask) -> (emb fp32 [Ipad, E], pre fp32 [Ipad, E]) ; keeps x_final for backward."""
        S, H = self.S, self.H
        pk = self._pk
        M = pad_to(pk['Mtok'], 256) if pk is not None else pad_to(n_items * S, 256)
        self._twin.clear()
        key_mask = self._buf('kmask', n_items, S, torch.float32)          # filled by a4r_embed_ln from the mask half of the rows
        x = self._buf('xa', M, H, self.T)
        # training with kept block inputs (LoRA, trainable q / k / v, the parallel placement): the producer writes a layer's input straight into
        # that layer's kept buffer instead of the layer copying it (one [M, H] copy per layer)
        direct_xin = lambda j: (saved is not None and j < len(self.bert_blocks) and 'xin' in saved[j] and 'y2' not in saved[j]
                                and tuple(saved[j]['xin'].shape) == (M, H))
        if direct_xin(0):
            x = saved[0]['xin']
        keep = self.train_emb and saved is not None
        word = self.emb_word
        if self.prompt_n:                          # refresh the learned rows, point the first n ids of every title at them
            n, V = self.prompt_n, self.prompt_V
            word = self.emb_word_ext
            word[V:V + n].copy_(self.prompt_param.detach())
            news = news.clone()
            red = torch.arange(V, V + n, device=news.device).expand(news.shape[0], n)
            if self.roberta:                       # RoBERTa derives the position ids from the ORIGINAL ids: a redirected pad stays a pad
                red = torch.where(news[:, :n] == self.pad_id, -red - 1, red)      # (a4r_embed_ln: negative id = row -(id + 1), counted as pad)
            news[:, :n] = red
        x_emb = self._buf('x_unpacked', pad_to(n_items * S, 256), H, self.T) if pk is not None else x
        L.embed_ln(news, word, self.emb_pos, self.emb_type0, self.emb_ln.gamma, self.emb_ln.beta, self.emb_ln.eps,
                   x_emb, n_items, S, roberta=self.roberta, pad_id=self.pad_id,
                   drop_p=self.p_hidden if train else 0.0, drop_site=999, drop_seed=seed,
                   pre_out=self._buf('emb_pre', M, H, self.T) if keep else None,
                   stats_out=self._buf('emb_st', M, 2, torch.float32) if keep else None, key_mask_out=key_mask)
        if pk is not None:                         # the attended tokens of every title, title after title; rows behind them zero; no key mask: no pad rows exist
            L.rows_idx_copy(x_emb, x, pk['map'], pk['Mtok'])
            if M > pk['Mtok']:
                L.zero(x[pk['Mtok']:M])
            key_mask = None
        self._news = news
        other = self._buf('xb', M, H, self.T)
        xa_buf = self._buf('xa', M, H, self.T)     # (the transient buffer, also when the embedding went straight into layer 0's kept input)
        Ip = pad_to(n_items, 128)
        cls = self._buf('cls', Ip, H, self.T)
        last = len(self.bert_blocks) - 1
        x8 = None                                  # fp8 encoder: (e4m3 rows, row scales) of x, handed from layer to layer
        for i, blk in enumerate(self.bert_blocks):
            cmode = self.cls_only and i == last
            bufs = saved[i] if saved is not None else self._block_bufs('bert.shared', blk, M, True, Mc=Ip if cmode else None)
            w8 = self.fp8 and i != last
            if cmode:
                self._block_forward(blk, x, key_mask, n_items, M, bufs, train, seed, cls, cls_rows=Ip, x8=x8)
            elif 'y2' in bufs:                     # the layer's output is kept per layer (backward reads it, _vskip): no ping-pong
                x8 = self._block_forward(blk, x, key_mask, n_items, M, bufs, train, seed, bufs['y2'], x8=x8, want8=w8)
                x = bufs['y2']
            else:
                out = other if x.data_ptr() != other.data_ptr() else xa_buf       # a transient buffer that is not the current input (x may be a kept y2)
                if direct_xin(i + 1) and not (self.fp8 and i != last):
                    out = saved[i + 1]['xin']
                x8 = self._block_forward(blk, x, key_mask, n_items, M, bufs, train, seed, out, x8=x8, want8=w8)
                x = out
            if not cmode:
                if (i + 1) in self.bert_klist and i != last:
                    self._buf(f'khs{i + 1}', M, H, self.T).copy_(x)      # hidden_states[i + 1], read by a K-Adapter below
        if self.bert_kads:
            self._kad_chain_forward(x, n_items, M, Ip, train, seed, cls, saved is not None)
        elif not self.cls_only:
            self._cls_gather(x, cls, n_items, S)
        emb = self._buf('emb', Ip, self.E, torch.float32)
        pre = self._buf('embpre', Ip, self.E, torch.float32)
        L.gemm_nt(cls, self.fc_w, emb, bias=self.fc_b, C2=pre, act=L.ACT_GELU, M=Ip)
        return emb, pre, key_mask, M

    def _user_forward(self, xin, log_mask, B, train, seed, saved):
        """xin fp32 [Mu, E] rows (b, t) ; log_mask [B, T] -> prec [Mu, E]."""
        E, Tn = self.E, self.Lseq - 1
        Mu = pad_to(B * Tn, 128)
        x = self._buf('sx_a', Mu, E, torch.float32)
        st0 = self._buf('sst0', Mu, 2, torch.float32)
        L.ln_fwd(xin, self.sas_ln0.gamma, self.sas_ln0.beta, self.sas_ln0.eps, x, st0, M=Mu, add=self.pos_emb[:Tn],
                 drop_p=self.p_sas if train else 0.0, drop_site=4000, drop_seed=seed)
        if self._sas_fused_ok():                       # one launch per block; backward recomputes from the block inputs kept here
            xs = [x] + [self._buf(f'sas.fused{j}', Mu, E, torch.float32) for j in range(len(self.sas_blocks))]
            for j, blk in enumerate(self.sas_blocks):
                L.sasrec_block(self._sas_desc(blk, seed, False), xs[j], log_mask, xs[j + 1], B, Tn, train)
            self._sas_xs = xs
            return xs[-1], Mu
        other = self._buf('sx_b', Mu, E, torch.float32)
        keep = saved is not None
        if self.sas_kads and keep and (getattr(self, '_kad_saved_s', None) is None or self._kad_saved_Mu != Mu):
            self._kad_saved_s = [self._kad_bufs(k, Mu) for k in self.sas_kads]
            self._kad_saved_Mu = Mu
        last = None
        for j, blk in enumerate(self.sas_blocks):
            if self.sas_kads:                      # model.py:573-583: the adapter reads (block input + previous adapter output)
                kad = self.sas_kads[j]
                sv = self._kad_saved_s[j] if keep else None
                fus = sv['fus'] if sv is not None else self._buf(kad.tag + '.fus', Mu, E, torch.float32)
                fus.copy_(x)
                if last is not None:
                    fus.add_(last)
                out = self._buf(kad.tag + '.out', Mu, E, torch.float32)
                self._kad_forward(kad, fus, B, Mu, train, seed, out, sv)
                last = out
            bufs = saved[j] if saved is not None else self._block_bufs('sas.shared', blk, Mu, True)
            self._block_forward(blk, x, log_mask, B, Mu, bufs, train, seed, other)
            x, other = other, x
        if self.sas_kads:
            cat = self._buf('skcat', Mu, 2 * E, torch.float32)
            cat[:, :E].copy_(x)
            cat[:, E:].copy_(last)
            y = self._buf('sk_y', Mu, E, torch.float32)
            L.gemm_nt(cat, self.d_com2.w, y, bias=self.d_com2.b, M=Mu)
            x = y
        return x, Mu

    def _pre_forward(self, n_items):
        """torch-side preparation of a step that must sit BEFORE its first kernel (image tower: the ViT-MAE masking order)."""

    # ------------------------------------------------------------------ public: inference entry points
    def _stack_attrs(self, news, n):
        """[n, sum 2 S_a] (ids | mask per attribute) -> [n_attr * n, 2 S] int64, attribute-major, every attribute right-padded to the longest one
        with pad id / mask 0 (data movement only)."""
        S = self.S0
        out = torch.zeros(self.n_attr * n, 2 * S, dtype=torch.int64, device=news.device)
        if self.roberta:
            out[:, :S] = self.pad_id
        for k, (st, nw) in enumerate(self.attrs):
            out[k * n:(k + 1) * n, :nw] = news[:n, st:st + nw]
            out[k * n:(k + 1) * n, S:S + nw] = news[:n, st + nw:st + 2 * nw]
        return out

    _ADD_DESC_BYTES = 32                       # sizeof(a4r_add_desc_t)

    def _attr_table(self, key, src, dst_off_rows, n, into_rows):
        """descriptor table of one a4r_unpack_add launch that adds src blocks scaled by 1 / n_attr (cached per buffer address and item count)"""
        cache = self.__dict__.setdefault('_attr_tabs', {})
        k = (key, src.data_ptr(), n)
        if k not in cache:
            if len(cache) >= 64:                   # ragged histories change n almost every step: keep the table count (and the H2D copies' garbage) bounded
                cache.clear()
            E, a = self.E, 1.0 / self.n_attr
            ents = [L.AddDesc(src.data_ptr() + (0 if into_rows else j * n * E * 4), (j * n * E if into_rows else 0), n, E, E, a) for j in range(self.n_attr)]
            cache[k] = L.desc_table(ents, self.dev)
        return cache[k]

    def _attr_mean(self, emb_all, n):
        """emb [pad(n), E] = mean over the attributes of emb_all [n_attr * n, E] (encoders.py:96-98)"""
        emb = self._buf('emb_mean', pad_to(n, 128), self.E, torch.float32)
        L.zero(emb)
        tab = self._attr_table('fwd', emb_all, 0, n, False)
        for j in range(self.n_attr):               # one launch per attribute: the descriptors of a launch run concurrently and these share their target
            L.unpack_add(emb, tab[j * self._ADD_DESC_BYTES:], 1, n * self.E)
        return emb

    def _attr_spread(self, d_emb, n):
        """its backward: every attribute's vector receives d_emb / n_attr"""
        Ip = pad_to(self.n_attr * n, 128)
        d_all = self._buf('d_emb_attr', Ip, self.E, torch.float32)
        L.zero(d_all)
        L.unpack_add(d_all, self._attr_table('bwd', d_emb, 0, n, True), self.n_attr, n * self.E)
        return d_all, Ip

    @torch.no_grad()
    def encode_items(self, news):
        L.require_gpu(news)
        self._set_S(self.S0)
        self._pk = None                            # (inference: the rectangular layout)
        n = news.shape[0]
        news = news.contiguous()
        if news.dtype != torch.int64:
            news = news.long()
        self.pack_trainables()
        if self.n_attr > 1:
            emb, _, _, _ = self._encode(self._stack_attrs(news, n), self.n_attr * n, False, 0, None)
            return self._attr_mean(emb, n)[:n].clone()
        emb, _, _, _ = self._encode(news, n, False, 0, None)
        return emb[:n].clone()

    @torch.no_grad()
    def user_encode(self, input_embs, log_mask):
        L.require_gpu(input_embs, log_mask)
        B, Tn, E = input_embs.shape
        assert Tn == self.Lseq - 1 and E == self.E
        Mu = pad_to(B * Tn, 128)
        xin = self._buf('sxin', Mu, E, torch.float32)
        xin[:B * Tn].copy_(input_embs.reshape(B * Tn, E).float())
        lm = log_mask.float().contiguous()
        self.pack_trainables()
        out, _ = self._user_forward(xin, lm, B, False, 0, None)
        return out[:B * Tn].view(B, Tn, E).clone()

    # ------------------------------------------------------------------ public: training step
    # Item slots whose embeddings no line of the model ever reads (SURVEY 8a "results-neutral savings"): per user the batch carries L positives
    # and L negatives in the order p0 n0 p1 n1 ... (dataset.py:24-49 -> view(-1, 2S), run.py:591).  Model.forward (model.py:48-70) scores
    # neg[:, :-1]: the LAST negative (which the dataset fills with the pad item) is dropped -- slot 2L - 1.  ModelCPC.forward (:113-135) scores the
    # last position only: of the negatives just n[L-2] is read.  Those items are not encoded (1 of 42 / 20 of 42 per user): the kept rows are
    # gathered in front of the item tower, their embeddings scattered back into the full [B, L, 2] layout (the other rows stay zero: never read,
    # their gradients are exactly zero) and the gradient rows gathered again for the tower's backward.  Every kept item sees the same arithmetic
    # as before (rows are independent through the tower); with dropout ON the masks are indexed by the compact row, i.e. another, equally valid
    # draw.  A4R_SKIP_UNUSED_ITEMS=0: encode all 2L slots (A/B runs).

    def _kept_rows(self, B):
        """Rows of the compact item batch: n_c; None when every slot is encoded.  CPC drops 20 of 42 items: always worth it (RoBERTa + Pfeiffer + CPC
        16.97 -> 9.97 ms per step).  SASRec drops 1 of 42: the tower's large launches cost whole ROUNDS of 256-row tiles on the CUs, so 2.4 % fewer
        rows pay only where they remove a round (ViT-B/16 at 8 users: 259 -> 253 row panels = 4 -> 3 rounds of the H-wide GEMMs, 29.2 -> 26.9 ms); at
        the headline's 158 -> 154 panels (2 rounds either way) the step was 1.5 % SLOWER with the gather / scatter launches added: all slots stay."""
        if _os.environ.get('A4R_SKIP_UNUSED_ITEMS', '1') == '0' or self.Lseq < 3:
            return None
        if self.arch == 'cpc':
            return B * (self.Lseq + 1)
        n_c = B * (2 * self.Lseq - 1)
        if _os.environ.get('A4R_SKIP_UNUSED_ITEMS') == '2':           # (A/B runs: compact whatever the round count)
            return n_c
        ncu = 256
        if torch.device(self.dev).type == 'cuda':
            ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        ntn = max(1, self.H // 256)
        rounds = lambda n: -(-(pad_to(n * self.S, 256) // 256 * ntn) // ncu)
        return n_c if rounds(n_c) < rounds(B * 2 * self.Lseq) else None

    host_log_mask = None           # set by Model.forward when run.py hands log_mask over on the host
    host_lens = None               # likewise: numpy int32 [items]: attended tokens per title, when every mask of the batch is a prefix (else None)
    host_max_tokens = None         # likewise: the longest title (tokens with attention mask 1) among the batch's items, read from the host copy

    def _set_S(self, S):
        """Tokens per item of the text tower for what runs next (pad tokens inside a title never reach the CLS output -- masked as keys,
        their own rows are not read).  When every title of a training batch is shorter than the data's title length, the step runs on the first S
        tokens only; inference and the next step start from the full length again."""
        if S != self.S and type(self) is TransRecEngine:
            self.S = S
            for b in self.bert_blocks:
                b.S = S

    def _kept_index(self, hm, B):
        """Ragged histories: BuildTrainDataset pads a short user's positives AND negatives with item 0 (dataset.py:24-49) and neither
        forward reads those slots -- position p's input is masked out of every valid query's attention, pad positions are outside the loss.  From the
        HOST copy of log_mask [B, L-1]: rows (int32, ascending) of the slots that ARE read -- positive p when it is an input (mask[p]) or a target
        (mask[p-1]), negative p when mask[p] (CPC: the inputs, the last target, negative L-2).  None: every slot is read (the static rule decides)."""
        import numpy as np
        lm = (hm.detach().reshape(B, self.Lseq - 1).numpy() != 0)
        Ls, T = self.Lseq, self.Lseq - 1
        pos = np.zeros((B, Ls), bool)
        neg = np.zeros((B, Ls), bool)
        pos[:, :T] |= lm
        if self.arch == 'cpc':
            pos[:, Ls - 1] = True
            neg[:, T - 1] = True
        else:
            pos[:, 1:] |= lm
            neg[:, :T] = lm
        if self.sas_kads:              # KAdapterBlock attends with an all-ones mask (modules.py:175-185): pad positions' inputs DO reach valid outputs
            pos[:] = True
        need = np.stack([pos, neg], 2).reshape(-1)
        n_c = int(need.sum())
        n_static = self._kept_rows(B)
        if n_c == 0 or n_c >= (n_static if n_static is not None else B * 2 * Ls):
            return None
        rows = np.nonzero(need)[0].astype(np.int32)
        return torch.from_numpy(rows).to(self.dev, non_blocking=True), n_c, rows

    def _slots_copy(self, full, comp, B, to_compact, idx=None):
        """Between the full slot layout `full` [>= B*2L, W] and the compact one `comp` [>= n_c, W] (fp32 views, W % 4 == 0): strided row copies only.
        SASRec: the first 2L - 1 slots of every user (a [B, (2L-1) W] block of the [B, 2L W] view); CPC: all positives (every second row), then n[L-2].
        idx = (device int32 rows, n_c): the indexed form for ragged batches (a4r_rows_idx_copy)."""
        if idx is not None:
            if to_compact:
                L.rows_idx_copy(full, comp, idx[0], idx[1])
            else:
                L.rows_idx_copy(comp, full, idx[0], idx[1], scatter=True)
            return
        Ls, W = self.Lseq, full.shape[1]
        if self.arch == 'cpc':
            f2 = full[2 * Ls - 3:]
            if to_compact:
                L.gather_rows(full, comp, B * Ls, 2)
                L.gather_rows(f2, comp[B * Ls:], B, 2 * Ls)
            else:
                L.scatter_rows(comp, full, B * Ls, 2)
                L.scatter_rows(comp[B * Ls:], f2, B, 2 * Ls)
            return
        k = (2 * Ls - 1) * W
        fv = full[:B * 2 * Ls].view(B, 2 * Ls * W)[:, :k]
        cv = comp[:B * (2 * Ls - 1)].view(B, k)
        if to_compact:
            L.gather_rows(fv, cv, B, 1)
        else:
            L.gather_rows(cv, fv, B, 1)

    def train_forward(self, sample_items, log_mask):
        """sample_items [B*L*2, 2S] int64, log_mask [B, L-1] -> loss (0-d fp32 device tensor)."""
        L.require_gpu(sample_items, log_mask)
        train = self.model.training
        n_full = sample_items.shape[0]
        B = n_full // (2 * self.Lseq)
        assert B * 2 * self.Lseq == n_full and log_mask.shape == (B, self.Lseq - 1)
        news = sample_items.contiguous()
        lm = log_mask.float().contiguous()
        n_items = n_full
        n_c = self._kept_rows(B)
        kidx = None
        hm, self.host_log_mask = self.host_log_mask, None
        if hm is not None and not hm.is_cuda and tuple(hm.shape) == (B, self.Lseq - 1) and _os.environ.get('A4R_SKIP_UNUSED_ITEMS', '1') != '0':
            kidx = self._kept_index(hm, B)
            if kidx is not None:
                n_c = kidx[1]
        row_bytes = news[0].numel() * news.element_size() if n_full else 0
        noise = getattr(self, 'next_noise', None)  # (ViT-MAE: explicit masking noise comes per item of the FULL layout)
        if noise is not None and n_c is not None:
            noise = noise.to(self.dev, torch.float32).contiguous()
            if noise.dim() != 2 or noise.shape[0] != n_full or (noise.shape[1] * 4) % 16:
                n_c = None
        if n_c is not None and row_bytes % 16 == 0:
            if noise is not None:
                nzc = self._buf('noise_c', n_c, noise.shape[1], torch.float32)
                self._slots_copy(noise, nzc, B, True, kidx)
                self.next_noise = nzc
            src = news.view(n_full, -1).view(torch.float32)
            comp = self._buf('items_c', n_c, src.shape[1], torch.float32)
            self._slots_copy(src, comp, B, True, kidx)
            news = comp.view(news.dtype).view((n_c,) + tuple(news.shape[1:]))
            n_items = n_c
        else:
            n_c, kidx = None, None
        # short titles: the step's token count (even: 16-byte row pieces).  Not with K-Adapter blocks on the text tower (KAdapterBlock attends with an
        # all-ones mask: pad tokens reach the CLS row there) and not with a soft prompt.
        hmt, self.host_max_tokens = self.host_max_tokens, None
        S_step = self.S0
        if (hmt is not None and type(self) is TransRecEngine and not self.bert_kads and not self.prompt_n and self.S0 % 2 == 0 and news.dim() == 2
                and news.dtype == torch.int64 and news.shape[1] == 2 * self.S0 and _os.environ.get('A4R_SKIP_UNUSED_ITEMS', '1') != '0'):
            S_step = min(self.S0, max(2, (int(hmt) + 1) // 2 * 2))
        self._set_S(S_step)
        # packed titles: every item runs on ITS OWN attended tokens (title lengths from the host copy of the rows, Model.forward), not on the batch's
        # longest title.  Needs prefix masks (a left-padded tokenizer or a mask with holes keeps the rectangular form), the MFMA attention kernels
        # (head dim 32 / 64) and a frozen embedding (its backward is written for the rectangular layout).  A4R_PACK_TITLES=0: off (A/B runs).
        hl, self.host_lens = self.host_lens, None
        self._pk = None
        if (hl is not None and hmt is not None and S_step <= self.S0 and type(self) is TransRecEngine and not self.bert_kads and not self.prompt_n
                and not self.train_emb and not self.fp8 and self.bert_blocks and self.bert_blocks[0].dh in (32, 64) and len(hl) == n_full
                and not getattr(self.bert_blocks[0], 'long', False)             # (the long attention kernels take no offsets)
                and _os.environ.get('A4R_PACK_TITLES', '1') != '0' and _os.environ.get('A4R_SKIP_UNUSED_ITEMS', '1') != '0'):
            import numpy as np
            if kidx is not None:
                rows_h = kidx[2]
            elif n_c is not None:                  # the static compaction's row order (_slots_copy)
                Ls = self.Lseq
                if self.arch == 'cpc':
                    rows_h = np.concatenate([np.arange(0, B * 2 * Ls, 2), np.arange(B) * 2 * Ls + 2 * Ls - 3])
                else:
                    rows_h = (np.arange(B)[:, None] * 2 * Ls + np.arange(2 * Ls - 1)[None, :]).reshape(-1)
            else:
                rows_h = None
            lens = np.minimum(hl if rows_h is None else hl[rows_h], S_step).astype(np.int64)
            Mtok = int(lens.sum())
            if len(lens) == n_items and Mtok < n_items * S_step - 255:        # (worth a gather only when it removes at least a row panel)
                off = np.zeros(n_items + 1, np.int64)
                np.cumsum(lens, out=off[1:])
                rmap = np.arange(Mtok, dtype=np.int64) + np.repeat(np.arange(n_items, dtype=np.int64) * S_step - off[:-1], lens)
                self._pk = dict(off=torch.from_numpy(off.astype(np.int32)).to(self.dev, non_blocking=True),
                                map=torch.from_numpy(rmap.astype(np.int32)).to(self.dev, non_blocking=True), Mtok=Mtok, n=n_items)
        if S_step < self.S0:                       # rows [ids(S0) | mask(S0)] -> [ids(S) | mask(S)]
            src = news.view(torch.float32)
            dst = self._buf('items_t', n_items, 4 * S_step, torch.float32)
            L.gather_rows(src[:, :2 * S_step], dst[:, :2 * S_step], n_items, 1)
            L.gather_rows(src[:, 2 * self.S0:2 * self.S0 + 2 * S_step], dst[:, 2 * S_step:], n_items, 1)
            news = dst.view(torch.int64)
        n_enc = n_items                            # rows the tower encodes: the kept items, times the news attributes
        if self.n_attr > 1:
            news = self._stack_attrs(news, n_items)
            n_enc = self.n_attr * n_items
        self._pre_forward(n_enc)
        self.pack_trainables()
        self.step_count += 1
        seed = (self.seed * 1000003 + self.step_count) & 0xFFFFFFFFFFFF
        M = pad_to(self._pk['Mtok'], 256) if self._pk is not None else pad_to(n_enc * self.S, 256)
        Mu = pad_to(B * (self.Lseq - 1), 128)
        Ipc = pad_to(n_enc, 128)
        if self._saved_bert is None or self._saved_M != M or self._saved_Mu != Mu or getattr(self, '_saved_Ip', Ipc) != Ipc:
            self._saved_Ip = Ipc                   # (the item count changes from step to step when pad slots are left out)
            nb = len(self.bert_blocks)
            self._saved_bert = [self._block_bufs(f'bert.{i}', b, M, False, Mc=Ipc if (self.cls_only and i == nb - 1) else None)
                                for i, b in enumerate(self.bert_blocks)]
            self._saved_sas = [self._block_bufs(f'sas.{j}', b, Mu, False) for j, b in enumerate(self.sas_blocks)]
            self._saved_M, self._saved_Mu = M, Mu
        saved_b, saved_s = self._saved_bert, self._saved_sas
        emb, pre, key_mask, M = self._encode(news, n_enc, train, seed, saved_b)
        if self.n_attr > 1:
            emb = self._attr_mean(emb, n_items)
        if n_c is not None:                        # back to the [B, L, 2] slot layout the head indexes
            emb_c = emb
            emb = self._buf('emb_full', pad_to(n_full, 128), self.E, torch.float32)
            L.zero(emb)
            self._slots_copy(emb, emb_c, B, False, kidx)
        xin = self._buf('sxin', Mu, self.E, torch.float32)
        L.take_inputs(emb, xin, B, self.Lseq, self.E)
        prec, Mu = self._user_forward(xin, lm, B, train, seed, saved_s)
        pos = self._buf('pos', B, self.Lseq - 1, torch.float32)
        neg = self._buf('neg', B, self.Lseq - 1, torch.float32)
        ws = self._buf('lossws', 1, 4, torch.float32)
        L.zero(ws)
        L.score_bce_fwd(emb, prec, lm, pos, neg, ws, B, self.Lseq, self.E, self.arch == 'cpc')
        self._ctx = dict(B=B, n_items=n_items, n_enc=n_enc, n_full=n_full, kidx=kidx, S=self.S, pk=self._pk, M=M, Mu=Mu, seed=seed, train=train, lm=lm, key_mask=key_mask, emb=emb, pre=pre,
                         prec=prec, xin=xin, pos=pos, neg=neg, ws=ws, saved_b=saved_b, saved_s=saved_s)
        return ws[0, 0].clone()

    def scores(self):
        """(pos_score, neg_score) [B, L-1] of the last train_forward (parity instrumentation)."""
        c = self._ctx
        return c['pos'].clone(), c['neg'].clone()

    OVERLAP_ALLREDUCE = bool(int(_os.environ.get('A4R_OVERLAP_ALLREDUCE', '1')))

    def _grad_chunks(self):
        """Flat-buffer ranges in the order backward FINISHES them: the user encoder (+ anything outside the item encoder's layers
        that its backward completes), then the item encoder's layers last to first, then the rest (item head, embeddings, LayerNorms
        outside the layers).  A chunk holding a gradient that only reaches the flat buffer at the very end (a zero-padded scratch
        corner: the reference's 16-wide SASRec adapters) is marked late and goes out after the flush.  None (= the single all-reduce)
        for Compacter, whose PHM chain fills every adapter gradient at the end."""
        if getattr(self, '_chunks', 0) != 0:
            return self._chunks
        self._chunks = None
        if self._virtual or not self.OVERLAP_ALLREDUCE:
            return None
        late_ids = {id(p) for _, p, _, _, _ in self._corners}
        import re
        user, layers, rest, late = [], {}, [], set()
        for n, p in zip(self.trainable_names, self.trainable_params):
            o, k = self.offsets[id(p)]
            m = re.search(r'encoder\.layer\.(\d+)\.', n)
            if 'user_encoder' in n:
                user.append((o, o + k))
                key = 'user'
            elif m and ('bert_encoder' in n or 'cv_encoder' in n):
                layers.setdefault(int(m.group(1)), []).append((o, o + k))
                key = int(m.group(1))
            else:
                rest.append((o, o + k))
                key = 'rest'
            if id(p) in late_ids:
                late.add(key)
        span = lambda r: (min(a for a, _ in r), max(b for _, b in r)) if r else None
        out = dict(user=span(user), layers={i: span(r) for i, r in layers.items()}, rest=span(rest), late=late)
        spans = sorted([s_ for s_ in [out['user'], out['rest']] + list(out['layers'].values()) if s_])
        if any(a[1] > b[0] for a, b in zip(spans, spans[1:])):          # interleaved groups: no contiguous ranges to exchange separately
            return None
        self._chunks = out
        return out

    def _exchange(self, what):
        """Start the all-reduce of one finished chunk (no-op outside an overlapped data-parallel backward)."""
        ex = getattr(self, '_exch', None)
        if ex is None:
            return
        ddp, ch, done, final = ex
        rng = ch['layers'].get(what) if isinstance(what, int) else ch[what]
        if rng is None or what in done or (what in ch['late'] and not final):
            return
        done.add(what)
        self._wgrad_join()                                     # the chunk's side-stream weight gradients have landed
        ddp.launch_(self.flat_g, rng[0], rng[1])

    def backward_bound(self, grad_out):
        """Backward of the public path (loss.backward(), run.py:599) once FusedAdam owns the flat buffers: every p.grad IS a view
        of flat_g, so the kernels accumulate straight into it (no per-parameter gradient list, no AccumulateGrad adds) and the
        data-parallel exchange is ONE all-reduce of flat_g.  After optimizer.zero_grad() flat_g is known to be zero and is the
        kernels' target; otherwise (gradient accumulation) the step goes to the scratch buffer and is added."""
        ddp = getattr(self.model, '_a4r_ddp', None)
        if self._flat_clean:
            self._flat_clean = False
            ch = self._grad_chunks() if ddp is not None else None
            if ch is not None:            # chunks are exchanged as backward finishes them, overlapped with the layers still to come
                self._exch = (ddp, ch, set(), False)
                try:
                    self.train_backward(grad_out, into_flat_grad=True)
                    self._exch = self._exch[:3] + (True,)         # the scratch corners are flushed: late chunks may go
                    for what in ['user'] + sorted(ch['layers'], reverse=True) + ['rest']:
                        self._exchange(what)                      # whatever the hooks did not reach (e.g. layers a frozen-input backward skips)
                finally:
                    self._exch = None
                    ddp.wait_all()                                # also after an exception: nothing stays in flight into the next step
                return
            self.train_backward(grad_out, into_flat_grad=True)
            if ddp is not None:                                   # DDP semantics: gradients averaged over ranks (run.py:503,599)
                ddp.average_(self.flat_g)
            return
        self.train_backward(grad_out, into_flat_grad=False, as_list=False)
        if ddp is not None:
            ddp.average_(self.flat_gs)
        self.flat_g.add_(self.flat_gs)

    def train_backward(self, grad_out=None, into_flat_grad=False, as_list=True):
        """Native backward of the last train_forward.  into_flat_grad: accumulate straight into the flat
        gradient buffer (fused path); else into the scratch buffer, returned as per-parameter gradients for autograd to
        accumulate (as_list).  grad_out: d(loss) as a 0-d fp32 DEVICE tensor (read by the head kernel, no host sync)."""
        c = self._ctx
        if c is None:
            raise RuntimeError('train_backward without train_forward')
        self._ctx = None
        target = self.flat_g if into_flat_grad else self.flat_gs
        if not into_flat_grad:
            target.zero_()
        self._grad_target = target
        self._zero_scratch()
        if grad_out is not None:
            grad_out = grad_out.detach().to(torch.float32).reshape(1)
        B, n_items, M, Mu, seed = c['B'], c['n_items'], c['M'], c['Mu'], c['seed']
        self._set_S(c.get('S', self.S))            # (the token count the forward ran on)
        self._pk = c.get('pk')
        n_full = c.get('n_full', n_items)          # the head works on the full slot layout, the item tower on the kept rows (train_forward)
        E, Tn = self.E, self.Lseq - 1
        train = c['train']
        Ip = pad_to(n_full, 128)
        d_prec = self._buf_tail0('d_prec', Mu, E, torch.float32, B * Tn)     # score_bce_bwd writes the real rows only
        d_emb = self._buf_tail0('d_emb', Ip, E, torch.float32, n_full)
        L.score_bce_bwd(c['emb'], c['prec'], c['lm'], c['pos'], c['neg'], c['ws'], 1.0, d_prec, d_emb, B, self.Lseq, E, self.arch == 'cpc',
                        scale_dev=grad_out)
        # SASRec blocks, last to first
        dx = d_prec
        dlast = None
        if self.sas_kads:                          # com_dense2 backward: [d block-chain output ; d last adapter output]
            self._dense_wgrad(self.d_com2, d_prec, self._buf('skcat', Mu, 2 * E, torch.float32), Mu)
            dcat = self._buf('sk_dcat', Mu, 2 * E, torch.float32)
            L.gemm_nt(d_prec, self.d_com2.wT, dcat, M=Mu)
            dx = self._buf('sk_dx', Mu, E, torch.float32)
            dx.copy_(dcat[:, :E])
            dlast = self._buf('sk_dlast', Mu, E, torch.float32)
            dlast.copy_(dcat[:, E:])
        fused = self._sas_fused_ok()
        if fused:        # the one-launch block kernels write rows < B * Tn only; ln_bwd below sums all Mu rows into sas_ln0's gamma / beta gradients
            pp = [self._buf_tail0('sdx_a', Mu, E, torch.float32, B * Tn), self._buf_tail0('sdx_b', Mu, E, torch.float32, B * Tn)]
        else:
            pp = [self._buf('sdx_a', Mu, E, torch.float32), self._buf('sdx_b', Mu, E, torch.float32)]
        for k, j in enumerate(range(len(self.sas_blocks) - 1, -1, -1)):
            if fused:
                L.sasrec_block(self._sas_desc(self.sas_blocks[j], seed, True), self._sas_xs[j], c['lm'], pp[k % 2], B, Tn, train, dy=dx)
            else:
                self._block_backward(self.sas_blocks[j], dx, c['lm'], B, Mu, c['saved_s'][j], train, seed, pp[k % 2])
            dx = pp[k % 2]
            if self.sas_kads:
                kad = self.sas_kads[j]
                d_fus = self._buf(kad.tag + '.dfus', Mu, E, torch.float32)
                self._kad_backward(kad, dlast, B, Mu, train, seed, d_fus, self._kad_saved_s[j])
                dx.add_(d_fus)
                dlast = d_fus
        d_in = self._buf('sd_in', Mu, E, torch.float32)
        st0 = self._buf('sst0', Mu, 2, torch.float32)
        gg = lambda f: f() if f is not None else None
        L.ln_bwd(dx, c['xin'], st0, self.sas_ln0.gamma, d_in, M=Mu, add=self.pos_emb[:Tn],
                 dgamma=gg(self.sas_ln0.g_gamma), dbeta=gg(self.sas_ln0.g_beta),
                 drop_p=self.p_sas if train else 0.0, drop_site=4000, drop_seed=seed)
        if self.g_pos_emb is not None:             # nn.Embedding(position) backward: the same rows are read by every user
            self.g_pos_emb()[:Tn].add_(d_in[:B * Tn].view(B, Tn, E).sum(0))
        L.emb_grad_add_inputs(d_in, d_emb, B, self.Lseq, E)
        self._exchange('user')
        if n_full != n_items:
            Ip = pad_to(n_items, 128)
            d_emb_c = self._buf_tail0('d_emb_c', Ip, E, torch.float32, n_items)
            self._slots_copy(d_emb, d_emb_c, B, True, c.get('kidx'))
            d_emb = d_emb_c
        if self.n_attr > 1:                        # the mean over the news attributes: every attribute's rows get d_emb / n_attr
            d_emb, Ip = self._attr_spread(d_emb, n_items)
            c = dict(c, n_items=c['n_enc'])
        self._items_backward(c, d_emb, Ip)
        self._wgrad_join()
        self._flush_corners()
        if self._virtual:
            self._virtual_backward()
        if into_flat_grad or not as_list:
            return None
        out = target.clone()
        ddp = getattr(self.model, '_a4r_ddp', None)
        if ddp is not None:                                   # DDP semantics: gradients averaged over ranks (run.py:503,599)
            ddp.average_(out)
        return [out[o:o + n].view(p.shape) for p, (o, n) in ((p, self.offsets[id(p)]) for p in self.trainable_params)]

    def _items_backward(self, c, d_emb, Ip):
        """d_emb [Ip, E] fp32 -> item head, encoder layers last to first (writes the adapter gradients)."""
        n_items, M, seed, train, E = c['n_items'], c['M'], c['seed'], c['train'], self.E
        # item head backward: GELU, fc dgrad, scatter to the CLS rows
        d_pre = self._buf('d_pre', Ip, E, torch.float32)
        L.act_bwd_f32(d_emb, c['pre'], d_pre, L.ACT_GELU)
        dcls = self._buf('dcls', Ip, self.H, self.T)
        L.gemm_nt(d_pre, self.fc_wT32, dcls, M=Ip)
        if self.d_fc.trainable:                    # item head fc (encoders.py:44,56) under --fine_tune_to all
            if self.d_fc.g_w is not None:
                L.gemm_tn(d_pre if self.T == torch.float32 else d_pre.to(self.T), self._buf('cls', Ip, self.H, self.T), self.d_fc.g_w(), M=Ip)
            if self.d_fc.g_b is not None:
                L.colsum(d_pre, self.d_fc.g_b(), M=Ip)
        dxb = self._buf('dx_a', M, self.H, self.T)
        d_hs = {}
        if self.bert_kads:
            d_hs = self._kad_chain_backward(dcls, n_items, M, Ip, train, seed, dxb)
            if not (self.bert_trains or self.train_emb):
                return                             # frozen backbone: nothing trainable lies upstream of its activations
        elif not self.cls_only:
            self._cls_scatter_fill(dcls, dxb, n_items, self.S, M)
        spare = self._buf('dx_b', M, self.H, self.T)
        last = len(self.bert_blocks) - 1
        for i in range(last, -1, -1):
            blk = self.bert_blocks[i]
            if i < last:
                self._exchange(i + 1)              # layer i + 1 is finished: its gradients go out while the layers below run
            if (i + 1) in d_hs:
                dxb.add_(d_hs[i + 1])              # hidden_states[i + 1] also fed a K-Adapter
            if self.cls_only and i == last:
                self._block_backward(blk, dcls, c['key_mask'], n_items, M, c['saved_b'][i], train, seed,
                                     spare if blk.need_dx else None, cls_rows=Ip)
            else:
                self._block_backward(blk, dxb, c['key_mask'], n_items, M, c['saved_b'][i], train, seed, spare if blk.need_dx else None)
            dxb, spare = spare, dxb
        if self.train_emb:                         # HF BertEmbeddings backward: LayerNorm (through its dropout mask), then the three tables
            gg = lambda f: f() if f is not None else None
            dpre = spare
            L.ln_bwd(dxb, self._buf('emb_pre', M, self.H, self.T), self._buf('emb_st', M, 2, torch.float32), self.emb_ln.gamma, dpre, M=M,
                     dgamma=gg(self.emb_ln.g_gamma), dbeta=gg(self.emb_ln.g_beta),
                     drop_p=self.p_hidden if train else 0.0, drop_site=999, drop_seed=seed)
            if self.g_prompt is not None:          # d learned_embedding[t] = sum over items of the token-t rows
                self.g_prompt().add_(dpre[:n_items * self.S].view(n_items, self.S, self.H)[:, :self.prompt_n].float().sum(0))
            if self.g_word is not None or self.g_pos is not None:
                L.embed_bwd(self._news, dpre, gg(self.g_word), gg(self.g_pos), n_items, self.S, roberta=self.roberta, pad_id=self.pad_id)
            if self.g_type is not None:
                L.colsum(dpre, self.g_type()[0], M=M)

    def _virtual_backward(self):
        """Compacter: the gradients of the effective matrices (the zero-padded scratch of the weight-gradient GEMMs) chained into
        (phm_rule, W_left, W_right) by one a4r_phm_bwd launch."""
        L.phm_bwd(self.flat_p, self._phm_tab, self._phm_n, self._grad_target)
