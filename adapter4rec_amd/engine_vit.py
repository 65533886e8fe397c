"""Image item tower on the MI355X-native engine: frozen ViT-B/16 (HF ViTForImageClassification) or ViT-MAE encoder with
injected Houlsby / Compacter adapters or LoRA on q, v -> the same SASRec/CPC user encoder, head, backward, flat-gradient
and Adam machinery as the text path (engine.py).

Replaces, behind ``Model.forward`` / ``Vit_Encoder`` / ``MAE_Encoder`` (reference: Downstream/CV/model/model.py:54-77,
encoders.py:8-32; HF transformers==4.20.1 modeling_vit.py / modeling_vit_mae.py, third party), per PRE-LN layer
    x1 = x  + A1(dense(attn(LN_before(x))))          A1 = VITAdaptedSelfOutput (model.py:182-195)
    x2 = x1 + A2(dense(gelu(dense(LN_after(x1)))))   A2 = VITAdaptedOutput     (model.py:198-212)
with A(h) = fc_up(act(fc_down(h))) + h (Houlsby) or without the inner residual (Compacter, model.py:432-462), then
``layernorm`` -> CLS -> classifier / cv_proj -> GELU.

Launch sequence per layer (all C-ABI kernels of liba4r_hip.so; the residual adds ride in GEMM epilogues):
    ln_fwd | gemm(qkv) | attn_long_fwd | gemm(o [+R1=x]) [| gemm(down) | gemm(up, R1=h, R2=x)] | ln_fwd | gemm(fc1, GELU)
    | gemm(fc2 [+R1=x1]) [| gemm(down) | gemm(up, R1=h, R2=x1)]
The input side is a4r_patchify (im2col + uint8 normalise; ViT-MAE: only the kept 25 % of the patches) -> one GEMM with
the Conv2d weight -> a4r_vit_assemble (cls, position rows).  Hidden / attention dropout are 0 in the ViT configs the
reference loads (google/vit-base-patch16-224, facebook/vit-mae-base) and are required to be 0 here.
"""
import math

import torch

from . import _lib as L
from .engine import TransRecEngine, _Adapter, _Block, _Dense, _LN, _Lora, pad_to
from .cv.vit import vit_geometry
from .model.modules import AdapterBlock, HyperComplexAdapterBlock


class ViTRecEngine(TransRecEngine):
    WGRAD_SIDE_OK = False           # the image tower's gradient buffers ping-pong: its weight gradients stay on the main stream

    # ------------------------------------------------------------------ build
    def _build_item_tower(self):
        enc = self.model.cv_encoder
        net = enc.image_net
        self.mae = not hasattr(net, 'vit')
        core = net if self.mae else net.vit
        g = self.geo = vit_geometry(net)
        H, nh = g['hidden_size'], g['num_attention_heads']
        self.H, self.F = H, g['intermediate_size']
        self.P, self.R, self.C = g['patch_size'], g['image_size'], g['num_channels']
        if H % 64 or self.F % 64 or H // nh != 64 or self.P % 8 or (self.C * self.P * self.P) % 64:
            raise NotImplementedError(f'ViT geometry H={H} F={self.F} heads={nh} patch={self.P}')
        if g['hidden_act'] != 'gelu':
            raise NotImplementedError(f"hidden_act {g['hidden_act']}")
        if g['hidden_dropout_prob'] or g['attention_probs_dropout_prob']:
            raise NotImplementedError('the image tower assumes the ViT configs the reference loads: dropout probabilities 0')
        self.NP = (self.R // self.P) ** 2
        self.n_keep = int(self.NP * (1 - g['mask_ratio'])) if self.mae else self.NP       # HF: int(seq_length * (1 - mask_ratio))
        emb = core.embeddings
        self.n_prompt, self.g_prompt = 0, None
        if type(emb).__name__ == 'SoftPrompt':       # model.py:512-535: learned tokens appended after cls + patches
            self.prompt_param = emb.Prompt_Tokens
            self.n_prompt = int(emb.n_tokens)
            self.g_prompt = self.grad_view(emb.Prompt_Tokens)
            emb = emb.wte
        self.S = self.n_keep + 1 + self.n_prompt
        if self.S > 256:
            raise NotImplementedError(f'{self.S} tokens per image (attention kernel: <= 256)')
        proj = emb.patch_embeddings.projection
        # Conv2d weight [H, C, P, P] is [H, C*P*P] in memory: the flat fp32 master is packed as that matrix when trainable
        self.d_patch = _Dense(self, proj.weight, proj.bias, self.T, view2d=(H, self.C * self.P * self.P))
        self.patch_w, self.patch_b = self.d_patch.w, self.d_patch.b
        tab = lambda p: p.data if p.requires_grad else self._f32(p)
        self.cls_tok = tab(emb.cls_token).reshape(H)
        self.pos_tab = tab(emb.position_embeddings).reshape(self.NP + 1, H)
        self.g_cls, self.g_postab = self.grad_view(emb.cls_token), self.grad_view(emb.position_embeddings)
        self.train_emb = self.d_patch.trainable or self.g_cls is not None or self.g_postab is not None or self.g_prompt is not None
        if self.g_prompt is not None and self.mae and self.train_emb:
            raise NotImplementedError('soft prompt on a ViT-MAE tower with a trainable embedding side is not wired')
        self.res24 = False                                # (the default 24-bit stream is a post-LN text-tower form: the pre-LN image tower stores its residual stream v itself)
        if self.res32:
            raise NotImplementedError('--residual_dtype fp32 is wired for the text tower (post-LN sub-layers on the one-launch adapter kernels); the pre-LN '
                                      'image tower stores its residual stream v itself in the compute dtype')
        self.next_noise = None
        self.bert_blocks = []
        kmod, enc_mod = None, core.encoder
        if type(enc_mod).__name__ == 'VITKAdaptedCVModel':       # model.py:374-404: the wrapper sits where vit.encoder was
            kmod, enc_mod = enc_mod, enc_mod.vit_encoder
        for i, layer in enumerate(enc_mod.layer):
            b = _Block()
            att = layer.attention.attention
            for lin in (att.query, att.key, att.value):
                if type(lin).__name__ not in ('LoRALinear', 'Linear'):
                    raise NotImplementedError(f'projection module {type(lin).__name__}')
            b.lora = _Lora.for_block((att.query, att.key, att.value), H, self, self.T)
            b.H, b.F, b.nh, b.dh, b.S = H, self.F, nh, 64, self.S
            b.scale = 1.0 / math.sqrt(64)
            b.wqkv = torch.zeros(3 * H, H, dtype=self.T, device=self.dev)
            b.wqkvT = torch.zeros(H, 3 * H, dtype=self.T, device=self.dev)
            b.bqkv = torch.zeros(3 * H, dtype=torch.float32, device=self.dev)
            self._pack_lora_bias(b, H)
            b.qkv = tuple(None if type(lin).__name__ == 'LoRALinear' else
                          _Dense(self, lin.weight, lin.bias, self.T, b.wqkv[sl * H:(sl + 1) * H], b.wqkvT[:, sl * H:(sl + 1) * H],
                                 b.bqkv[sl * H:(sl + 1) * H])
                          for sl, lin in enumerate((att.query, att.key, att.value)))
            d1, b.ad1 = self._vit_so(layer.attention.output)
            d2, b.ad2 = self._vit_so(layer.output)
            b.d_o = _Dense(self, d1.weight, d1.bias, self.T)
            b.d_i = _Dense(self, layer.intermediate.dense.weight, layer.intermediate.dense.bias, self.T)
            b.d_o2 = _Dense(self, d2.weight, d2.bias, self.T)
            b.wo, b.woT, b.bo = b.d_o.w, b.d_o.wT, b.d_o.b
            b.wi, b.wiT, b.bi = b.d_i.w, b.d_i.wT, b.d_i.b
            b.wo2, b.wo2T, b.bo2 = b.d_o2.w, b.d_o2.wT, b.d_o2.b
            b.train_dense = any(d is not None and d.trainable for d in b.qkv + (b.d_o, b.d_i, b.d_o2))
            b.lnA, b.lnB = _LN(layer.layernorm_before, self), _LN(layer.layernorm_after, self)
            b.need_dx = i > 0 or self.train_emb
            b.T = self.T
            self.bert_blocks.append(b)
        self.vit_ln = _LN(core.layernorm, self)
        fc = enc.cv_proj if self.mae else net.classifier
        if self.E % 64 or fc.out_features != self.E:
            raise NotImplementedError('item head: classifier / cv_proj with embedding_dim % 64 == 0')
        self.d_fc = _Dense(self, fc.weight, fc.bias, self.T)
        self.fc_w, self.fc_b = self.d_fc.w, self.d_fc.b
        self.fc_wT32 = torch.zeros(fc.in_features, fc.out_features, dtype=torch.float32, device=self.dev)
        if fc.weight.requires_grad:
            self.add_pack(fc.weight, self.fc_wT32, True)
        else:
            self.fc_wT32.copy_(fc.weight.detach().t().float())
        self.cls_only = bool(getattr(self.args, 'cls_only_last', True))      # last layer: only the CLS rows go past attention
        self.bert_trains = any(p.requires_grad for p in enc_mod.parameters())
        self.bert_kads, self.bert_klist, self.d_com = [], [], None
        if kmod is not None:
            nb = len(self.bert_blocks)
            self.bert_klist = [int(k) for k in kmod.k_adapter_num_list]
            if any(k < 1 or k > nb for k in self.bert_klist):
                raise ValueError(f'--k_adapter_bert_list {self.bert_klist} outside 1..{nb}')
            if self.fp8:
                raise NotImplementedError('K-Adapter on the fp8 image tower')
            self.bert_kads = [self._make_kadapter(a, H, self.S, self.T, 6000 + 64 * j) for j, a in enumerate(kmod.bert_adapter_list)]
            self.d_com = _Dense(self, kmod.com_dense.weight, kmod.com_dense.bias, self.T)
            self.cls_only = False                      # the adapters attend over all tokens of the last layer's output

    def _vit_so(self, mod):
        """(dense Linear, adapter or None) of a plain or wrapped ViTSelfOutput / ViTOutput."""
        if hasattr(mod, 'self_output'):
            ad = getattr(mod, 'adapter', None)
            pl = getattr(mod, 'placement', 'serial')
            if not isinstance(ad, (AdapterBlock, HyperComplexAdapterBlock)) or pl not in ('serial', 'parallel'):
                raise NotImplementedError(f'ViT adapter wrapper {type(mod).__name__}')
            a = _Adapter(ad, self.H, self, self.T)
            a.parallel = pl == 'parallel'
            return mod.self_output.dense, a
        return mod.dense, None

    # ------------------------------------------------------------------ buffers
    def _block_bufs(self, tag, blk, M, shared, Mc=None):
        if not hasattr(blk, 'lnA'):
            return super()._block_bufs(tag, blk, M, shared, Mc)
        pre = tag if not shared else tag.split('.')[0] + '.shared'
        T, H, F = blk.T, blk.H, blk.F
        d = {}
        if not shared:
            d['x0'] = self._buf(pre + '.x0', M, H, T)                      # the layer input (LN_before backward needs it)
        if (blk.lora or blk.train_dense) and not shared:
            d['n1'] = self._buf(pre + '.n1', M, H, T)
        d['sta'] = self._buf(pre + '.sta', M, 2, torch.float32)
        d['qkv'] = self._buf(pre + '.qkv', M, 3 * H, T)
        d['lse'] = self._buf(pre + '.lse', (M // blk.S + 1) * blk.nh * blk.S, 1, torch.float32)
        if not shared:
            d['ctx_o'] = self._buf(pre + '.ctx_o', M, H, T)                # the attention output: its backward takes delta = dO . O from it
        if Mc is not None:                   # last layer in CLS-only mode: everything after attention has Mc rows
            pre, M = pre + '.cls', Mc
        d['x1'] = self._buf(pre + '.x1', M, H, T)
        if blk.train_dense and not shared:       # inputs of the trainable Linears (weight gradients, --fine_tune_to all)
            d['ctx_s'] = self._buf(pre + '.ctx_s', M, H, T)
            d['n2_s'] = self._buf(pre + '.n2_s', M, H, T)
            d['u_s'] = self._buf(pre + '.u_s', M, F, T)
        d['stb'] = self._buf(pre + '.stb', M, 2, torch.float32)
        d['upre'] = self._buf(pre + '.upre', M, F, torch.uint8 if self._q8(blk) else T)
        for k, ad in (('1', blk.ad1), ('2', blk.ad2)):
            if ad is not None:
                d['h' + k] = self._buf(pre + '.h' + k, M, H, T)
                d['zp' + k] = self._buf(pre + '.zp' + k, M, ad.dp, T)
                d['z' + k] = self._buf(pre + '.z' + k, M, ad.dp, T)
        return d

    # ------------------------------------------------------------------ one pre-LN layer
    def _vit_sub_forward(self, ad, dense_in, w, bias, resid, bufs, k, M, out, scales=None):
        """scales = (scale_a, scale_b): dense_in / w are e4m3 operands (the fp8 encoder's FFN-down)."""
        sk = dict(scale_a=scales[0], scale_b=scales[1]) if scales is not None else {}
        if ad is None:
            L.gemm_nt(dense_in, w, out, bias=bias, R1=resid, M=M, **sk)
            return
        h, zp, z = bufs['h' + k], bufs['zp' + k], bufs['z' + k]
        if getattr(ad, 'parallel', False):    # model.py:165-179: dense(u) + x + [fc_up(act(fc_down(x))) + x], the adapter reads the sub-layer INPUT x
            L.gemm_nt(resid, ad.wd, z, bias=ad.bd, C2=zp, act=ad.act, M=M)
            L.gemm_nt(z, ad.wu, h, bias=ad.bu, R1=resid, R2=resid, M=M)
            L.gemm_nt(dense_in, w, out, bias=bias, R1=h, M=M, **sk)
            return
        L.gemm_nt(dense_in, w, h, bias=bias, M=M, **sk)
        L.gemm_nt(h, ad.wd, z, bias=ad.bd, C2=zp, act=ad.act, M=M)
        if ad.kind == 'compacter':            # no inner residual (Downstream/CV/model/modules.py HyperComplexAdapterBlock.forward)
            L.gemm_nt(z, ad.wu, out, bias=ad.bu, R1=resid, M=M)
        else:
            L.gemm_nt(z, ad.wu, out, bias=ad.bu, R1=h, R2=resid, M=M)

    def _vit_fuse(self, blk, ad, t, bwd=False):
        """The one-launch adapter + residual + LayerNorm kernels (a4r_adapter_fused.hip) serve the pre-LN tower too: v = the residual
        stream after the sub-layer, y = the NEXT LayerNorm's output (LN_after for the attention sub-layer, the next layer's LN_before
        for the FFN sub-layer).  Serial placement, bf16, bottleneck 64."""
        if ad is None or getattr(ad, 'parallel', False):
            return False
        return (self._fuse_bwd if bwd else self._fuse)(blk, ad, t)

    def _vit_block_forward(self, blk, x, n_items, M, bufs, x_out, cls_rows=None, n1_done=False, nxt=None):
        """cls_rows = Ip: after attention only token 0 of every image (all the head reads, encoders.py:22,32) is carried on:
        x_out is then [Ip, H].  Results-neutral: the other rows of the last layer's output are never consumed.
        n1_done: the previous layer's fused FFN-adapter kernel already wrote this layer's LN_before output and statistics.
        nxt = (next block, its buffers): this layer's FFN adapter is fused with the next layer's LN_before.  -> True when it was."""
        T, H = blk.T, blk.H
        fp8_qkv = self.fp8 and blk.wqkv8 is not None
        if n1_done:
            n1 = bufs['n1'] if 'n1' in bufs else self._buf('n1', M, H, T)
            if fp8_qkv:                                 # (the fused kernel of the layer below wrote the e4m3 row and its scale)
                n1q, n1s = self._buf('n1q', M, H, torch.uint8), self._buf('n1s', M, 1, torch.float32)
                L.gemm_nt(n1q, blk.wqkv8, bufs['qkv'], bias=blk.bqkv, M=M, scale_a=n1s, scale_b=blk.wqkv8s)
            else:
                L.gemm_nt(n1, blk.wqkv, bufs['qkv'], bias=blk.bqkv, M=M)
        elif fp8_qkv:      # LN_before emits the row as e4m3 + scale: the qkv GEMM runs on fp8 operands
            n1q, n1s = self._buf('n1q', M, H, torch.uint8), self._buf('n1s', M, 1, torch.float32)
            L.ln_fwd(x, blk.lnA.gamma, blk.lnA.beta, blk.lnA.eps, bufs.get('n1'), bufs['sta'], M=M, y8=n1q, ys=n1s)
            L.gemm_nt(n1q, blk.wqkv8, bufs['qkv'], bias=blk.bqkv, M=M, scale_a=n1s, scale_b=blk.wqkv8s)
        else:
            n1 = bufs['n1'] if 'n1' in bufs else self._buf('n1', M, H, T)
            L.ln_fwd(x, blk.lnA.gamma, blk.lnA.beta, blk.lnA.eps, n1, bufs['sta'], M=M)
            L.gemm_nt(n1, blk.wqkv, bufs['qkv'], bias=blk.bqkv, M=M)
        ctx = bufs['ctx_o'] if 'ctx_o' in bufs else self._buf('ctx', M, H, T)
        L.attn_long_fwd(bufs['qkv'], ctx, bufs['lse'], n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.scale)
        if cls_rows is not None:
            ctx_c, x_c = self._buf('ctx_c', cls_rows, H, T), self._buf('x_c', cls_rows, H, T)
            L.gather_rows(ctx, ctx_c, n_items, blk.S)
            L.gather_rows(x, x_c, n_items, blk.S)
            ctx, x, M = ctx_c, x_c, cls_rows
        if 'ctx_s' in bufs:
            L.gather_rows(ctx, bufs['ctx_s'], M, 1)
        fp8_fc1 = self.fp8 and blk.wi8 is not None and M % 256 == 0
        n2 = bufs['n2_s'] if 'n2_s' in bufs else self._buf('n2', M, H, T)
        fp8_o = self.fp8 and blk.wo8 is not None and M % 256 == 0 and 'ctx_s' not in bufs
        if self._vit_fuse(blk, blk.ad1, ctx):       # dense | adapter + residual + LN_after in one launch
            ad = blk.ad1
            h = bufs['h1']
            if fp8_o:                                   # the attention output as e4m3 + per-token scale (one pass), then the fp8 GEMM
                c8, cs = self._buf('ctx8', M, H, torch.uint8), self._buf('ctx8s', M, 1, torch.float32)
                L.quant_rows_fp8(ctx, c8, cs, M=M)
                L.gemm_nt(c8, blk.wo8, h, bias=blk.bo, M=M, scale_a=cs, scale_b=blk.wo8s)
            else:
                L.gemm_nt(ctx, blk.wo, h, bias=blk.bo, M=M)
            comp = ad.kind == 'compacter'
            n2q = n2s = None
            if fp8_fc1:                                 # LN_after leaves the kernel as e4m3 + scale (bf16 copy only if something trains on it)
                n2q, n2s = self._buf('n2q', M, H, torch.uint8), self._buf('n2s', M, 1, torch.float32)
            L.adapter_ln_fwd(h, x if comp else h, None if comp else x, ad.wd, ad.bd, ad.wu, ad.bu, blk.lnB.gamma, blk.lnB.beta, blk.lnB.eps,
                             ad.act, bufs['zp1'], bufs['z1'], bufs['x1'], n2 if (not fp8_fc1 or 'n2_s' in bufs) else None, bufs['stb'], M=M,
                             y8=n2q, ys=n2s, frag=ad.frag_f)
            n2_done = True
        else:
            self._vit_sub_forward(blk.ad1, ctx, blk.wo, blk.bo, x, bufs, '1', M, bufs['x1'])
            n2_done = False
        u = bufs['u_s'] if 'u_s' in bufs else self._buf('u', M, blk.F, T)
        c2d = 'q8' if self._q8(blk) else True
        fp8_ffn = fp8_fc1 and blk.wo28 is not None and 'u_s' not in bufs       # FFN-up writes u as e4m3, FFN-down (and both dgrads) run on e4m3
        w2, w2s = (blk.wo28, (self._const_rows('su', M, self.FP8_U_SCALE), blk.wo28s)) if fp8_ffn else (blk.wo2, None)
        if fp8_fc1:
            n2q, n2s = self._buf('n2q', M, H, torch.uint8), self._buf('n2s', M, 1, torch.float32)
            if not n2_done:
                L.ln_fwd(bufs['x1'], blk.lnB.gamma, blk.lnB.beta, blk.lnB.eps, bufs.get('n2_s'), bufs['stb'], M=M, y8=n2q, ys=n2s)
            if fp8_ffn:
                u = self._buf('u8', M, blk.F, torch.uint8)
                L.gemm_nt(n2q, blk.wi8, u, bias=blk.bi, C2=bufs['upre'], act=L.ACT_GELU, c2_deriv='q8', M=M, scale_a=n2s, scale_b=blk.wi8s,
                          c_fp8=1, c_scale=self.FP8_U_SCALE, q8_tiled=self._q8t(blk, M))
            else:
                L.gemm_nt(n2q, blk.wi8, u, bias=blk.bi, C2=bufs['upre'], act=L.ACT_GELU, c2_deriv=c2d, M=M, scale_a=n2s, scale_b=blk.wi8s, q8_tiled=self._q8t(blk, M))
        else:
            if not n2_done:
                L.ln_fwd(bufs['x1'], blk.lnB.gamma, blk.lnB.beta, blk.lnB.eps, n2, bufs['stb'], M=M)
            L.gemm_nt(n2, blk.wi, u, bias=blk.bi, C2=bufs['upre'], act=L.ACT_GELU, c2_deriv=c2d, M=M, q8_tiled=self._q8t(blk, M))
        if nxt is not None and cls_rows is None and self._vit_fuse(blk, blk.ad2, bufs['x1']):
            nb_, nbufs = nxt                            # dense | adapter + residual + the NEXT layer's LN_before
            ad = blk.ad2
            h = bufs['h2']
            if w2s is not None:
                L.gemm_nt(u, w2, h, bias=blk.bo2, M=M, scale_a=w2s[0], scale_b=w2s[1])
            else:
                L.gemm_nt(u, w2, h, bias=blk.bo2, M=M)
            comp = ad.kind == 'compacter'
            n1n = nbufs['n1'] if 'n1' in nbufs else self._buf('n1', M, H, T)
            n1q = n1s = None
            if self.fp8 and nb_.wqkv8 is not None:
                n1q, n1s = self._buf('n1q', M, H, torch.uint8), self._buf('n1s', M, 1, torch.float32)
            L.adapter_ln_fwd(h, bufs['x1'] if comp else h, None if comp else bufs['x1'], ad.wd, ad.bd, ad.wu, ad.bu,
                             nb_.lnA.gamma, nb_.lnA.beta, nb_.lnA.eps, ad.act, bufs['zp2'], bufs['z2'], x_out,
                             n1n if (n1q is None or 'n1' in nbufs) else None, nbufs['sta'], M=M, y8=n1q, ys=n1s, frag=ad.frag_f)
            return True
        self._vit_sub_forward(blk.ad2, u, w2, blk.bo2, bufs['x1'], bufs, '2', M, x_out, scales=w2s)
        return False

    def _vit_sub_backward(self, blk, ad, dy, bufs, k, M, x_in=None):
        """dy: gradient of the sub-layer output before the residual add -> (gradient of the dense output, gradient that goes to
        the residual stream: dy itself, or 2 dy + dzp Wd for the parallel form whose adapter reads the sub-layer input x_in)."""
        if ad is None:
            return dy, dy
        T, H = blk.T, blk.H
        h, zp, z = bufs['h' + k], bufs['zp' + k], bufs['z' + k]
        dzp = self._buf('dzp', M, ad.dp, T)
        L.gemm_nt(dy, ad.wuT, dzp, Pre=zp, dact=ad.act, M=M)
        dh = self._buf('dh' + k, M, H, T)
        if getattr(ad, 'parallel', False):
            L.gemm_nt(dzp, ad.wdT, dh, R1=dy, R2=dy, M=M)
            self._adapter_wgrads(ad, dy, z, dzp, x_in, M)
            if ad.g_bu is not None:
                L.colsum(dy, ad.g_bu(), M=M)
            return dy, dh
        if ad.kind == 'compacter':
            L.gemm_nt(dzp, ad.wdT, dh, M=M)
        else:
            L.gemm_nt(dzp, ad.wdT, dh, R1=dy, M=M)
        self._adapter_wgrads(ad, dy, z, dzp, h, M)
        if ad.g_bu is not None:
            L.colsum(dy, ad.g_bu(), M=M)
        return dh, dy

    def _vit_block_backward(self, blk, dx_out, n_items, M, bufs, dx_in, cls_rows=None, pre2=None, prev=None):
        """pre2 = (d_o, dres2): the FFN adapter's backward was already done by the NEXT layer (fused with its LN_before backward).
        prev = (previous block, its buffers): this layer's LN_before backward runs fused with the previous layer's FFN adapter
        backward -> returns that layer's pre2 (else None)."""
        T, H, F = blk.T, blk.H, blk.F
        gg = lambda f: f() if f is not None else None
        M_full = M
        if cls_rows is not None:
            M = cls_rows
        if pre2 is not None:
            d_o, dres2 = pre2
        else:
            d_o, dres2 = self._vit_sub_backward(blk, blk.ad2, dx_out, bufs, '2', M, x_in=bufs['x1'])
        self._dense_wgrad(blk.d_o2, d_o, bufs.get('u_s'), M)
        dn2 = self._buf('dn', M, H, T)
        if self.fp8 and blk.wo2T8 is not None and M % 256 == 0 and bufs['upre'].dtype == torch.uint8 and 'u_s' not in bufs:
            # frozen FFN, both dgrads on e4m3 operands with ONE scale per token row carried through the chain (_build_fp8):
            #   d_o -> e4m3 + row scale | du = (d_o W2) * gelu' leaves its GEMM as e4m3 with scale[m] * c_du | dn2 = du W1 (bf16 out)
            do8, dos = self._buf('do8', M, H, torch.uint8), self._buf('do8s', M, 1, torch.float32)
            L.quant_rows_fp8(d_o, do8, dos, M=M)
            du8, dus = self._buf('du8', M, F, torch.uint8), self._buf('du8s', M, 1, torch.float32)
            L.gemm_nt(do8, blk.wo2T8, du8, Pre=bufs['upre'], dact=L.DACT_MUL_Q8, M=M, scale_a=dos, scale_b=blk.wo2T8s,
                      c_fp8=2, c_scale=blk.c_du, c_scale_out=dus, q8_tiled=self._q8t(blk, M))
            L.gemm_nt(du8, blk.wiT8, dn2, M=M, scale_a=dus, scale_b=blk.wiT8s)
        else:
            du = self._buf('du', M, F, T)
            L.gemm_nt(d_o, blk.wo2T, du, Pre=bufs['upre'], dact=L.DACT_MUL_Q8 if self._q8(blk) else L.DACT_MUL, M=M, q8_tiled=self._q8t(blk, M))
            self._dense_wgrad(blk.d_i, du, bufs.get('n2_s'), M)
            L.gemm_nt(du, blk.wiT, dn2, M=M)
        dx1 = self._buf('dx1', M, H, T)
        if self._vit_fuse(blk, blk.ad1, dn2, bwd=True):
            # ONE launch: LN_after backward (+ the residual-branch gradient), dzp = (dx1 Wu) * act'(zp), da = dzp Wd [+ dx1]
            ad = blk.ad1
            dzp = self._buf('dzp', M, ad.dp, T)
            da = self._buf('dh1', M, H, T)
            bd = self._bd_target(ad)
            b2 = self._tn2_bias_ok(ad, dx1, M)          # (bias_total: db_up = colsum of the TOTAL dx1 this launch writes = the weight-gradient launch's X operand)
            L.adapter_ln_bwd(dn2, bufs['x1'], bufs['stb'], blk.lnB.gamma, dres2, bufs['zp1'], ad.act, ad.wuT, ad.wdT, ad.kind != 'compacter',
                             dx1, dzp, da, dgamma=gg(blk.lnB.g_gamma), dbeta=gg(blk.lnB.g_beta), dbias=None if b2 else gg(ad.g_bu), M=M,
                             dbd=None if b2 else bd, bias_total=True, frag=ad.frag_b)
            self._adapter_wgrads(ad, dx1, bufs['z1'], dzp, bufs['h1'], M, bd_done=bd is not None, bias_in_tn2=b2)
        else:
            L.ln_bwd(dn2, bufs['x1'], bufs['stb'], blk.lnB.gamma, dx1, M=M, dgamma=gg(blk.lnB.g_gamma), dbeta=gg(blk.lnB.g_beta), dres=dres2)
            da, dres1 = self._vit_sub_backward(blk, blk.ad1, dx1, bufs, '1', M)
            assert dres1 is dx1              # the parallel form exists at layer.output only (run_adapter.py:448-453)
        ln_a = blk.lnA.g_gamma is not None           # --finetune_layernorm: layer 0 still owes its LN_before gradients
        qkv_train = any(d is not None and d.trainable for d in blk.qkv)
        pend = []                                    # the attention output's weight gradient rides in the q / k / v launch below (engine.py)
        if qkv_train and cls_rows is None:
            pend = [(blk.d_o, da, bufs.get('ctx_s'))]
        else:
            self._dense_wgrad(blk.d_o, da, bufs.get('ctx_s'), M)
        if dx_in is None and not blk.lora and not ln_a and not qkv_train:
            return None
        dctx = self._buf('dctx_c' if cls_rows is not None else 'dctx', M, H, T)
        L.gemm_nt(da, blk.woT, dctx, M=M)
        if cls_rows is not None:             # back to token rows: the gradients live on the CLS rows only
            M = M_full
            full = self._buf('dctx', M, H, T)
            L.scatter_rows_fill(dctx, full, n_items, blk.S, M)          # CLS rows written, every other row zeroed, one pass
            dctx = full
            rfull = self._buf('dres_full', M, H, T)
            L.scatter_rows_fill(dx1, rfull, n_items, blk.S, M)
            dx1 = rfull
        dqkv = self._buf_tail0('dqkv', M, 3 * H, T, n_items * blk.S)       # attn_long_bwd writes the real token rows only
        ws = self._buf('attn_ws', bufs['lse'].shape[0], 1, torch.float32)
        L.attn_long_bwd(bufs['qkv'], bufs['ctx_o'], dctx, dqkv, bufs['lse'], ws, n_items, blk.S, blk.nh, blk.dh, 0, H, 2 * H, blk.scale)
        if blk.lora:
            self._lora_backward_all(blk, dqkv, bufs['n1'], M)
        self._dense_wgrads([(d, dqkv[:, sl * H:(sl + 1) * H], bufs.get('n1')) for sl, d in enumerate(blk.qkv)] + pend, M)
        if dx_in is not None or ln_a:
            if dx_in is None:
                dx_in = self._buf('dx_unused', M, H, T)
            dn1 = self._buf('dn', M, H, T)
            L.gemm_nt(dqkv, blk.wqkvT, dn1, M=M)
            if prev is not None and self._vit_fuse(prev[0], prev[0].ad2, dn1, bwd=True):
                # this layer's LN_before backward fused with the PREVIOUS layer's FFN-adapter backward (forward fused them too)
                pb, pbufs = prev
                ad = pb.ad2
                dzp = self._buf('dzp2', M, ad.dp, T)
                d_o_prev = self._buf('dh2', M, H, T)
                bd = self._bd_target(ad)
                b2 = self._tn2_bias_ok(ad, dx_in, M)
                L.adapter_ln_bwd(dn1, bufs['x0'], bufs['sta'], blk.lnA.gamma, dx1, pbufs['zp2'], ad.act, ad.wuT, ad.wdT, ad.kind != 'compacter',
                                 dx_in, dzp, d_o_prev, dgamma=gg(blk.lnA.g_gamma), dbeta=gg(blk.lnA.g_beta), dbias=None if b2 else gg(ad.g_bu), M=M,
                                 dbd=None if b2 else bd, bias_total=True, frag=ad.frag_b)
                self._adapter_wgrads(ad, dx_in, pbufs['z2'], dzp, pbufs['h2'], M, bd_done=bd is not None, bias_in_tn2=b2)
                return d_o_prev, dx_in
            L.ln_bwd(dn1, bufs['x0'], bufs['sta'], blk.lnA.gamma, dx_in, M=M, dgamma=gg(blk.lnA.g_gamma), dbeta=gg(blk.lnA.g_beta), dres=dx1)
        return None

    # ------------------------------------------------------------------ item tower
    def _keep_indices(self, n_items, noise):
        """ViT-MAE random masking (HF ViTMAEEmbeddings.random_masking): keep the n_keep patches of smallest noise, in
        argsort order.  noise None: drawn on the device (training); explicit noise: parity runs."""
        if not self.mae:
            return None
        keep = self._buf('mae_keep', n_items, self.n_keep, torch.int32)
        if noise is None:              # counter-hash noise per (seed, item, patch): a4r_mae_keep_indices draws and ranks it in one launch
            # stream = (dropout_seed, the engine's PERSISTED step counter, draws since that counter last moved, rank): a resumed run continues
            # the stream instead of replaying the first run's masks, and the ranks of a data-parallel job mask differently (as
            # DeviceTrainSampler.set_epoch does for the negatives)
            step = self.step_count + 1
            if getattr(self, '_mae_step', None) != step:
                self._mae_step, self._mae_draw = step, 0
            self._mae_draw += 1
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            mix = (self.seed * 1000003 + step) * 1000033 + self._mae_draw * 8191 + rank * 0x9E3779B97F4A7C15
            L.mae_keep_indices(keep, self.NP, None, seed=mix & 0xFFFFFFFFFFFF, site=4900)
        else:
            L.mae_keep_indices(keep, self.NP, noise.to(self.dev, torch.float32).contiguous())
        return keep

    _keep = None

    def _pre_forward(self, n_items):
        # the ViT-MAE masking order (torch rand + argsort) is drawn before the step's first HIP kernel, not between pack and patchify
        self._keep = self._keep_indices(n_items, self.next_noise)

    def _encode(self, images, n_items, train, seed, saved):
        S, H = self.S, self.H
        M = pad_to(n_items * S, 256)
        self._twin.clear()
        if images.dtype == torch.uint8:
            if images.shape[1:] != (self.R, self.R, self.C):
                raise ValueError(f'uint8 images must be [n, {self.R}, {self.R}, {self.C}] (HWC), got {tuple(images.shape)}')
        else:
            if images.shape[1:] != (self.C, self.R, self.R):
                raise ValueError(f'images must be [n, {self.C}, {self.R}, {self.R}], got {tuple(images.shape)} (no in-engine resize)')
            images = images.float()
        images = images.contiguous()
        keep = self._keep if self._keep is not None else self._keep_indices(n_items, self.next_noise)
        self._keep, self.next_noise = None, None
        self._keep_used = keep                      # (ViT-MAE with a trainable embedding side: the backward scatters by the same indices)
        Mp = pad_to(n_items * self.n_keep, 256)
        cols = self.C * self.P * self.P
        pat = self._buf('patches', Mp, cols, self.T)
        L.patchify(images, pat, self.P, keep)
        pe = self._buf('patch_emb', Mp, H, self.T)
        L.gemm_nt(pat, self.patch_w, pe, bias=self.patch_b, M=Mp)
        x = saved[0]['x0'] if saved is not None else self._buf('xa', M, H, self.T)
        L.vit_assemble(pe, self.cls_tok, self.pos_tab, x, n_items, self.n_keep, keep, tokens_out=S)
        if self.n_prompt:
            x[:n_items * S].view(n_items, S, H)[:, S - self.n_prompt:] = self.prompt_param.detach().to(self.T)
        other = self._buf('xb', M, H, self.T)
        nb = len(self.bert_blocks)
        Ip = pad_to(n_items, 128)
        cls = self._buf('cls', Ip, H, self.T)
        n1_done = False
        self._fused_next = {}                      # layer -> its FFN adapter ran fused with the next layer's LN_before (backward mirrors it)
        for i, blk in enumerate(self.bert_blocks):
            cmode = self.cls_only and i + 1 == nb
            if saved is not None:                 # training: layer i writes straight into layer i+1's saved input
                out = saved[i + 1]['x0'] if i + 1 < nb else (cls if cmode else self._buf('x_last', M, H, self.T))
                nxt = (self.bert_blocks[i + 1], saved[i + 1]) if i + 1 < nb else None
                n1_done = self._vit_block_forward(blk, x, n_items, M, saved[i], out, cls_rows=Ip if cmode else None, n1_done=n1_done, nxt=nxt)
                self._fused_next[i] = n1_done
                x = out
            else:                                 # inference: one transient buffer set, two ping-pong activations
                bufs = self._block_bufs('vit.shared', blk, M, True, Mc=Ip if cmode else None)
                nxt = (self.bert_blocks[i + 1], bufs) if i + 1 < nb else None
                n1_done = self._vit_block_forward(blk, x, n_items, M, bufs, cls if cmode else other, cls_rows=Ip if cmode else None,
                                                  n1_done=n1_done, nxt=nxt)
                x, other = (cls, other) if cmode else (other, x)
            if (i + 1) in self.bert_klist and i + 1 < nb:
                L.gather_rows(x, self._buf(f'khs{i + 1}', M, H, self.T), M, 1)      # hidden_states[i + 1], read by a K-Adapter below
        if self.bert_kads:       # model.py:389-404: adapters chained over the listed hidden states, com_dense([last ; adapter]) -> layernorm
            self._kad_chain_forward(x, n_items, M, Ip, train, seed, cls, saved is not None)
        elif not self.cls_only:
            L.gather_rows(x, cls, n_items, S)
        cln = self._buf('cls_n', Ip, H, self.T)
        self._cls_st = self._buf('cls_st', Ip, 2, torch.float32)
        L.ln_fwd(cls, self.vit_ln.gamma, self.vit_ln.beta, self.vit_ln.eps, cln, self._cls_st, M=Ip)
        emb = self._buf('emb', Ip, self.E, torch.float32)
        pre = self._buf('embpre', Ip, self.E, torch.float32)
        L.gemm_nt(cln, self.fc_w, emb, bias=self.fc_b, C2=pre, act=L.ACT_GELU, M=Ip)
        return emb, pre, None, M

    def _items_backward(self, c, d_emb, Ip):
        n_items, M, E, H = c['n_items'], c['M'], self.E, self.H
        gg = lambda f: f() if f is not None else None
        d_pre = self._buf('d_pre', Ip, E, torch.float32)
        L.act_bwd_f32(d_emb, c['pre'], d_pre, L.ACT_GELU)
        dcln = self._buf('dcls_n', Ip, H, self.T)
        L.gemm_nt(d_pre, self.fc_wT32, dcln, M=Ip)
        if self.d_fc.trainable:
            if self.d_fc.g_w is not None:
                L.gemm_tn(d_pre if self.T == torch.float32 else d_pre.to(self.T), self._buf('cls_n', Ip, H, self.T), self.d_fc.g_w(), M=Ip)
            if self.d_fc.g_b is not None:
                L.colsum(d_pre, self.d_fc.g_b(), M=Ip)
        dcls = self._buf('dcls', Ip, H, self.T)
        L.ln_bwd(dcln, self._buf('cls', Ip, H, self.T), self._cls_st, self.vit_ln.gamma, dcls, M=Ip,
                 dgamma=gg(self.vit_ln.g_gamma), dbeta=gg(self.vit_ln.g_beta))
        dxb = self._buf('dx_a', M, H, self.T)
        d_hs = {}
        if self.bert_kads:
            d_hs = self._kad_chain_backward(dcls, n_items, M, Ip, c['train'], c['seed'], dxb)
            if not (self.bert_trains or self.train_emb):
                return                             # frozen backbone: nothing trainable lies upstream of its activations
        elif not self.cls_only:
            L.scatter_rows_fill(dcls, dxb, n_items, self.S, M)
        spare = self._buf('dx_b', M, H, self.T)
        last = len(self.bert_blocks) - 1
        pre2 = None
        for i in range(last, -1, -1):
            blk = self.bert_blocks[i]
            if i < last:
                self._exchange(i + 1)              # layer i + 1 is finished: its gradients go out while the layers below run
            if (i + 1) in d_hs:
                dxb.add_(d_hs[i + 1])              # hidden_states[i + 1] also fed a K-Adapter
            prev = (self.bert_blocks[i - 1], c['saved_b'][i - 1]) if i > 0 and self._fused_next.get(i - 1) else None
            if self.cls_only and i == last:
                pre2 = self._vit_block_backward(blk, dcls, n_items, M, c['saved_b'][i], spare if blk.need_dx else None, cls_rows=Ip, prev=prev)
            else:
                pre2 = self._vit_block_backward(blk, dxb, n_items, M, c['saved_b'][i], spare if blk.need_dx else None, pre2=pre2, prev=prev)
            dxb, spare = spare, dxb
        if self.train_emb and self.mae:
            # ViTMAEEmbeddings backward (HF modeling_vit_mae.py, Pretraining/CV's shipped configuration: CV_model_load = 'mae', nothing frozen): token 0
            # = cls + pos[0]; token 1 + j = projection(kept patch j) + pos[1 + keep[j]].  The patch matrix of the forward holds exactly the kept
            # patches in token order, so dW = d(tokens 1..)^T patches with no scatter; position embeddings are a fixed sin-cos table in HF
            # (requires_grad False) -- if a caller did make them trainable their rows are reached through the keep indices.
            S, nk = self.S, self.n_keep
            d3 = dxb[:n_items * S].view(n_items, S, H)
            if self.g_cls is not None:
                self.g_cls().view(H).add_(d3[:, 0].float().sum(0))
            if self.g_postab is not None:
                gp = self.g_postab().view(self.NP + 1, H)
                gp[0].add_(d3[:, 0].float().sum(0))
                gp.index_add_(0, (self._keep_used[:n_items].long() + 1).reshape(-1), d3[:, 1:1 + nk].float().reshape(n_items * nk, H))
            if self.d_patch.trainable:
                Mp = pad_to(n_items * nk, 256)
                dpe = self._buf('d_patch_emb', Mp, H, self.T)
                dpe[:n_items * nk].copy_(d3[:, 1:1 + nk].reshape(n_items * nk, H))
                dpe[n_items * nk:].zero_()
                self._dense_wgrad(self.d_patch, dpe, self._buf('patches', Mp, self.C * self.P * self.P, self.T), Mp)
        elif self.train_emb:                     # ViTEmbeddings backward: token 0 -> cls + pos[0]; token 1 + j -> patch projection + pos[1 + j]
            S, NP = self.S, self.NP
            d3 = dxb[:n_items * S].view(n_items, S, H)
            if self.g_prompt is not None:
                self.g_prompt().view(self.n_prompt, H).add_(d3[:, S - self.n_prompt:].float().sum(0))
                d3 = d3[:, :S - self.n_prompt]
            if self.g_postab is not None:
                self.g_postab().view(NP + 1, H).add_(d3.float().sum(0))
            if self.g_cls is not None:
                self.g_cls().view(H).add_(d3[:, 0].float().sum(0))
            if self.d_patch.trainable:
                Mp = pad_to(n_items * NP, 256)
                dpe = self._buf('d_patch_emb', Mp, H, self.T)
                dpe[:n_items * NP].copy_(d3[:, 1:].reshape(n_items * NP, H))
                dpe[n_items * NP:].zero_()
                self._dense_wgrad(self.d_patch, dpe, self._buf('patches', Mp, self.C * self.P * self.P, self.T), Mp)

    # ------------------------------------------------------------------ public: inference
    @torch.no_grad()
    def encode_items(self, images, noise=None):
        L.require_gpu(images)
        n = images.shape[0]
        self.next_noise = noise
        self.pack_trainables()
        emb, _, _, _ = self._encode(images, n, False, 0, None)
        return emb[:n].clone()
