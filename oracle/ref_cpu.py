"""CPU oracle for the adapter-tuned TransRec hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``adapter4rec_amd``) never calls into ``oracle/`` and raises when
its HIP library is missing.

It is a functional fp32 restatement (plain torch on CPU, explicit formulas, no
``nn.Module``, no HuggingFace, no reference imports) of what the reference computes on
its training / eval path, operating on a flat ``{name: tensor}`` dict that uses the
reference's own ``state_dict`` key names.  Every function cites the reference lines
it follows (paths relative to ``/root/reference``).  Gradients of the restatement are
taken with torch autograd on the CPU.

Parity pin: ``tests/golden/*.npz`` were produced by ``tools/gen_golden.py`` by
importing the reference (``Downstream/Text/model``, ``data_utils``) in the build
container; ``tests/test_oracle_golden.py`` checks this file against them.
LoRA (third-party ``loralib==0.1.1``, absent from the reference tree and the image)
is restated from its published semantics (W x + b + (alpha / r) B A x, alpha = 1) and
pinned ALGEBRAICALLY through the reference's own numbers: with W = W_base - B A / r the
layer is the reference's plain Linear at W_base, so ``lora_linear`` must reproduce the
imported reference's forward and dA = B^T dW / r, dB = dW A^T / r must follow from the
reference's own dL/dW (``tests/golden/lora_pin_{text,image}.npz``, written by
``tools/gen_golden_r4.py``; ``test_lora_pinned_through_merged_weights``).  What stays
unpinned is only loralib's initialisation and its alpha default, which no run captures.
"""
import math
import random

import numpy as np
import torch

BERT = 'bert_encoder.text_encoders.title.bert_model.'
FC = 'bert_encoder.text_encoders.title.fc.'
UE = 'user_encoder.transformer_encoder.'

DEFAULT_CFG = dict(
    arch='sasrec',              # 'sasrec' | 'cpc'            Downstream/Text/run.py:360-363
    encoder='bert',             # 'bert' | 'roberta'          run.py:289-300
    bert_heads=12, bert_ln_eps=1e-12, pad_token_id=0,
    adapter_type='houslby',     # sic, parameters.py:67
    adapter_activation='RELU',  # parameters.py:64
    is_serial='True',           # parameters.py:66
    sasrec_heads=2, max_seq_len=20, embedding_dim=64, num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'],
    lora_r_bert=64, lora_r_sasrec=16,
)


# --------------------------------------------------------------------------- training-mode dropout (optional)
def _drop(cfg, kind, site, x, p_key, **kw):
    """cfg['drop'] = an oracle.dropout_masks.DropoutStream (tests only): multiply by the SAME keep mask the HIP kernels regenerate from
    (seed, site, element) at this site of the reference (torch.nn.Dropout, train mode).  Absent: eval mode, no dropout (the fixtures' mode)."""
    d = cfg.get('drop')
    if d is None or site is None:
        return x
    return x * d.mask(kind, site, x, float(cfg.get(p_key, 0.0)), **kw)


# --------------------------------------------------------------------------- math
def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gelu_new(x):  # transformers.activations "gelu_new" (tanh form), modules.py:220
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def adapter_act(name):
    """modules.py:121-124 (AdapterBlock: 'GELU' else ReLU); :143-148 (Pfeiffer)."""
    if name == 'GELU':
        return gelu_erf
    if name == 'leaky_relu':
        return lambda x: torch.where(x > 0, x, 0.01 * x)
    return lambda x: torch.clamp(x, min=0)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


# --------------------------------------------------------------------------- PHM (compacter)
def phm_matrix(phm_rule, w_left, w_right):
    """layers.py:10-22,150-160 + kronecker.py:23-34: H = sum_i kron(rule[i], W_left[i] @ W_right[i])."""
    w = torch.bmm(w_left, w_right)                       # [n, in/n, out/n]
    n, a, c = phm_rule.shape
    _, k, p = w.shape
    kron = (phm_rule[:, :, None, :, None] * w[:, None, :, None, :]).reshape(n, a * k, c * p)
    return kron.sum(0)                                   # [in, out]


def phm_linear(sd, prefix, x, phm_rule):
    h = phm_matrix(phm_rule, sd[prefix + 'W_left'], sd[prefix + 'W_right'])
    return x @ h + sd[prefix + 'b']


# --------------------------------------------------------------------------- adapters
def houlsby_block(sd, p, h, cfg):
    """modules.py:116-134 AdapterBlock: fc_up(act(fc_down(h))) + h (its dropout is never applied)."""
    f = adapter_act(cfg['adapter_activation'])
    return linear(f(linear(h, sd[p + 'fc_down.weight'], sd[p + 'fc_down.bias'])),
                  sd[p + 'fc_up.weight'], sd[p + 'fc_up.bias']) + h


def pfeiffer_block(sd, p, h, cfg):
    """modules.py:137-158 AdapterPfeifferBlock: no inner residual."""
    f = adapter_act(cfg['adapter_activation'])
    return linear(f(linear(h, sd[p + 'fc_down.weight'], sd[p + 'fc_down.bias'])),
                  sd[p + 'fc_up.weight'], sd[p + 'fc_up.bias'])


def compacter_block(sd, p, h, cfg):
    """modules.py:209-252 HyperComplexAdapterBlock: up(gelu_new(down(h))), no residual."""
    rule = sd[cfg.get('phm_rule_key', 'phm_rule')]
    z = gelu_new(phm_linear(sd, p + 'down_sampler.', h, rule))
    return phm_linear(sd, p + 'up_sampler.', z, rule)


def lora_linear(sd, p, x, r):
    """loralib==0.1.1 lora.Linear (third party; pinned through merged weights, see the module docstring): W x + b + (x A^T B^T) * (alpha/r), alpha=1."""
    y = linear(x, sd[p + 'weight'], sd.get(p + 'bias'))
    if p + 'lora_A' in sd and r > 0:
        y = y + (x @ sd[p + 'lora_A'].t() @ sd[p + 'lora_B'].t()) * (1.0 / r)
    return y


# --------------------------------------------------------------------------- BERT / RoBERTa
def position_ids(ids, cfg):
    """HF BertEmbeddings: arange(S); RobertaEmbeddings: cumsum(ids != pad) * (ids != pad) + pad."""
    n, s = ids.shape
    if cfg['encoder'] == 'roberta':
        m = (ids != cfg['pad_token_id']).long()
        return torch.cumsum(m, 1) * m + cfg['pad_token_id']
    return torch.arange(s).unsqueeze(0).expand(n, s)


def bert_embed(sd, ids, cfg):
    e = BERT + 'embeddings.'
    if e + 'word_embeddings.learned_embedding' in sd:      # soft prompt, model/model.py:586-630 SoftEmbedding: the first n_tokens
        le = sd[e + 'word_embeddings.learned_embedding']    # word vectors of every title are REPLACED by the learned rows
        n = le.shape[0]
        w = torch.cat([le.unsqueeze(0).expand(ids.shape[0], -1, -1), sd[e + 'word_embeddings.wte.weight'][ids[:, n:]]], 1)
    else:
        w = sd[e + 'word_embeddings.weight'][ids]
    x = w + sd[e + 'position_embeddings.weight'][position_ids(ids, cfg)] \
        + sd[e + 'token_type_embeddings.weight'][0]
    x = layer_norm(x, sd[e + 'LayerNorm.weight'], sd[e + 'LayerNorm.bias'], cfg['bert_ln_eps'])
    return _drop(cfg, 'rows', 999, x, 'p_hidden')          # HF BertEmbeddings.forward: LayerNorm -> dropout


def bert_self_output(sd, p, hidden, inp, cfg, site=None, rows='rows'):
    """The (possibly wrapped) BertSelfOutput / BertOutput at prefix ``p``.

    plain HF:                dense -> dropout -> LN(h + inp)
    BertAdaptedSelfOutput    model/model.py:292-297   (Houlsby serial)
    BertAdaptedParallel...   model/model.py:265-270
    BertPfeifferAdapted...   model/model.py:321-329
    BertCompacterAdapted...  model/model.py:715-720
    """
    eps = cfg['bert_ln_eps']
    if p + 'self_output.dense.weight' not in sd:                       # un-adapted
        h = _drop(cfg, rows, site, linear(hidden, sd[p + 'dense.weight'], sd[p + 'dense.bias']), 'p_hidden')
        return layer_norm(h + inp, sd[p + 'LayerNorm.weight'], sd[p + 'LayerNorm.bias'], eps)
    so = p + 'self_output.'
    h = linear(hidden, sd[so + 'dense.weight'], sd[so + 'dense.bias'])
    h = _drop(cfg, rows, site, h, 'p_hidden')               # every wrapper: self_output.dropout right behind self_output.dense (model.py:266-268, 293-294, 322-323, 716-717)
    lw, lb = sd[so + 'LayerNorm.weight'], sd[so + 'LayerNorm.bias']
    if p + 'LN.weight' in sd:                                          # Pfeiffer
        r = h
        t = layer_norm(h + inp, lw, lb, eps)
        t = pfeiffer_block(sd, p + 'adapter.', t, cfg) + r
        return layer_norm(t + inp, sd[p + 'LN.weight'], sd[p + 'LN.bias'], 1e-6)
    if p + 'adapter.down_sampler.W_left' in sd:                        # Compacter
        return layer_norm(compacter_block(sd, p + 'adapter.', h, cfg) + inp, lw, lb, eps)
    if 'None' in cfg['is_serial']:                                     # parallel Houlsby
        return layer_norm(houlsby_block(sd, p + 'adapter.', inp, cfg) + h + inp, lw, lb, eps)
    return layer_norm(houlsby_block(sd, p + 'adapter.', h, cfg) + inp, lw, lb, eps)


def bert_attention(sd, p, x, key_mask, cfg, site=None):
    """HF BertSelfAttention (eager): softmax(QK^T/sqrt(dh) + (1-mask)*finfo.min) V."""
    n, s, hdim = x.shape
    nh = cfg['bert_heads']
    dh = hdim // nh
    q = lora_linear(sd, p + 'query.', x, cfg['lora_r_bert'])
    k = linear(x, sd[p + 'key.weight'], sd[p + 'key.bias'])
    v = lora_linear(sd, p + 'value.', x, cfg['lora_r_bert'])
    q, k, v = [t.view(n, s, nh, dh).transpose(1, 2) for t in (q, k, v)]
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    sc = sc + (1.0 - key_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    pr = _drop(cfg, 'attn_item', site, torch.softmax(sc, -1), 'p_attn', head_dim=dh)      # HF BertSelfAttention: dropout(attention_probs)
    return (pr @ v).transpose(1, 2).reshape(n, s, hdim)


def bert_encode(sd, ids, key_mask, cfg, return_all=False):
    x = bert_embed(sd, ids, cfg)
    outs = [x]
    i = 0
    while BERT + f'encoder.layer.{i}.attention.self.query.weight' in sd:
        lp = BERT + f'encoder.layer.{i}.'
        # (masks only: behind the LAST layer's attention the product keeps the CLS row of every item -- all the item head reads -- so its
        # two dense dropouts there are drawn on [items, H] rows; 'rows_cls' puts that mask on token 0 and leaves the other tokens alone;
        # cfg['drop_cls_only'] = engine.cls_only: False for the parallel form and the K-Adapter, which need every token row)
        rows = 'rows' if (BERT + f'encoder.layer.{i + 1}.attention.self.query.weight' in sd or return_all or not cfg.get('drop_cls_only', True)) else 'rows_cls'
        ctx = bert_attention(sd, lp + 'attention.self.', x, key_mask, cfg, site=16 * i)
        x1 = bert_self_output(sd, lp + 'attention.output.', ctx, x, cfg, site=16 * i + 1, rows=rows)
        u = gelu_erf(linear(x1, sd[lp + 'intermediate.dense.weight'], sd[lp + 'intermediate.dense.bias']))
        x = bert_self_output(sd, lp + 'output.', u, x1, cfg, site=16 * i + 2, rows=rows)
        outs.append(x)
        i += 1
    return outs if return_all else x


def text_encoder(sd, news, cfg, return_all=False):
    """model/encoders.py:48-57,89-99: ids || mask -> BERT -> fc(CLS) -> GELU.  With more than one news attribute (--news_attributes title,abstract,body,
    encoders.py:62-99) every attribute's [ids | mask] slice -- laid out title, abstract, body, the lengths of the absent ones 0 -- goes through the
    SAME encoder ('title': the only Text_Encoder the reference builds) and the item vector is the mean of the attributes' vectors."""
    attrs = [a for a in ('title', 'abstract', 'body') if a in cfg.get('news_attributes', ['title'])]
    if len(attrs) > 1:
        assert not return_all
        vecs, start = [], 0
        for a in ('title', 'abstract', 'body'):
            nw = cfg['num_words_' + a] if a in attrs else 0
            if nw:
                vecs.append(text_encoder(sd, news[:, start:start + 2 * nw], dict(cfg, news_attributes=['title'], num_words_title=nw)))
            start += 2 * nw
        return torch.stack(vecs, 1).mean(1)
    nw = cfg['num_words_' + attrs[0]] if attrs else cfg['num_words_title']      # (Text_Encoder halves whatever slice it is given, encoders.py:49)
    ids, mask = news[:, :nw], news[:, nw:2 * nw]
    if BERT + 'com_dense.weight' in sd:          # K-Adapter, model/model.py:523-559 BertKAdaptedBertModel wraps the backbone
        inner = {(BERT + k[len(BERT + 'bert_model.'):] if k.startswith(BERT + 'bert_model.') else k): v for k, v in sd.items()}
        hs = bert_encode(inner, ids, mask, cfg, return_all=True)
        ks = [int(i) + 1 for i in str(cfg['k_adapter_bert_list']).split(',')]
        last = 0
        for j, k in enumerate(ks):                # hidden_states[k] = output of layer k-1, chained through the adapters
            last = kadapter_block(sd, BERT + f'bert_adapter_list.{j}.', hs[k] + last, cfg['num_adapter_heads_bert'], cfg)
        out = linear(torch.cat([hs[-1], last], -1), sd[BERT + 'com_dense.weight'], sd[BERT + 'com_dense.bias'])
        emb = gelu_erf(linear(out[:, 0], sd[FC + 'weight'], sd[FC + 'bias']))
        return (emb, hs) if return_all else emb
    hs = bert_encode(sd, ids, mask, cfg, return_all=return_all)
    last = hs[-1] if return_all else hs
    emb = gelu_erf(linear(last[:, 0], sd[FC + 'weight'], sd[FC + 'bias']))
    return (emb, hs) if return_all else emb


def kadapter_block(sd, p, x, n_heads, cfg):
    """model/modules.py:161-206 KAdapterBlock: down_project -> 2 plain post-LN TransformerBlocks under an all-zero additive mask (no
    key mask, NOT causal) -> up_project, + input."""
    if cfg.get('drop') is not None:
        raise NotImplementedError('training-mode masks are not restated for the K-Adapter blocks')
    h = linear(x, sd[p + 'down_project.weight'], sd[p + 'down_project.bias'])
    c = dict(cfg, sasrec_heads=n_heads, lora_r_sasrec=0, is_serial='True')
    for j in range(2):
        h = sasrec_block(sd, p + f'transformer_blocks.{j}.', h, torch.zeros(()), c)
    return x + linear(h, sd[p + 'up_project.weight'], sd[p + 'up_project.bias'])


# --------------------------------------------------------------------------- SASRec user encoder
def _sasrec_names(sd, bp):
    tb = bp + 'transformer_block.' if bp + 'transformer_block.multi_head_attention.w_K.weight' in sd else bp
    return tb + 'multi_head_attention.', tb + 'feed_forward.'


def sasrec_block(sd, bp, x, add_mask, cfg, site=None):
    """modules.py:45-87 (plain block) and the wrappers model/model.py:341-376 (Houlsby),
    :388-423 (pfeiffer_ver2), :435-471 (pfeiffer), :474-520 (parallel), :659-693 (compacter)."""
    mha, ff = _sasrec_names(sd, bp)
    b, t, d = x.shape
    nh = cfg['sasrec_heads']
    dk = d // nh
    q = lora_linear(sd, mha + 'w_Q.', x, cfg['lora_r_sasrec'])
    k = linear(x, sd[mha + 'w_K.weight'])
    v = lora_linear(sd, mha + 'w_V.', x, cfg['lora_r_sasrec'])
    q, k, v = [z.view(b, t, nh, dk).transpose(1, 2) for z in (q, k, v)]
    pr = torch.softmax(q @ k.transpose(-1, -2) / (dk ** 0.5) + add_mask, -1)
    pr = _drop(cfg, 'attn_user', site, pr, 'p_sas')                                        # modules.py:40 SelfAttention.dropout
    h = linear((pr @ v).transpose(1, 2).reshape(b, t, d), sd[mha + 'fc.weight'])
    h = _drop(cfg, 'rows_user', None if site is None else site + 1, h, 'p_sas')           # modules.py:70 dropout(fc(x)); every wrapper keeps it there
    ln1 = (sd[mha + 'layer_norm.weight'], sd[mha + 'layer_norm.bias'], 1e-6)
    ln2 = (sd[ff + 'layer_norm.weight'], sd[ff + 'layer_norm.bias'], 1e-6)

    def ffn(z):
        o = linear(torch.clamp(linear(z, sd[ff + 'w_1.weight'], sd[ff + 'w_1.bias']), min=0),
                   sd[ff + 'w_2.weight'], sd[ff + 'w_2.bias'])
        return _drop(cfg, 'rows_user', None if site is None else site + 2, o, 'p_sas')     # modules.py:27 dropout(w_2(relu(w_1 x)))

    if bp + 'LN.weight' in sd:                                         # pfeiffer
        x1 = layer_norm(x + h, *ln1)
        h2 = ffn(x1)
        t2 = layer_norm(x1 + h2, *ln2)
        t2 = pfeiffer_block(sd, bp + 'adapter.', t2, cfg) + h2
        return layer_norm(t2 + x1, sd[bp + 'LN.weight'], sd[bp + 'LN.bias'], 1e-6)
    if bp + 'adapter1.down_sampler.W_left' in sd:                      # compacter
        x1 = layer_norm(x + compacter_block(sd, bp + 'adapter1.', h, cfg), *ln1)
        return layer_norm(x1 + compacter_block(sd, bp + 'adapter2.', ffn(x1), cfg), *ln2)
    if bp + 'adapter1.fc_down.weight' in sd:
        if bp + 'adapter2.fc_down.weight' not in sd:                   # pfeiffer_ver2
            x1 = layer_norm(x + houlsby_block(sd, bp + 'adapter1.', h, cfg), *ln1)
            return layer_norm(x1 + ffn(x1), *ln2)
        if 'None' in cfg['is_serial']:                                 # parallel
            x1 = layer_norm(houlsby_block(sd, bp + 'adapter1.', x, cfg) + x + h, *ln1)
            return layer_norm(houlsby_block(sd, bp + 'adapter2.', x1, cfg) + x1 + ffn(x1), *ln2)
        x1 = layer_norm(x + houlsby_block(sd, bp + 'adapter1.', h, cfg), *ln1)
        return layer_norm(x1 + houlsby_block(sd, bp + 'adapter2.', ffn(x1), cfg), *ln2)
    x1 = layer_norm(x + h, *ln1)                                       # un-adapted
    return layer_norm(x1 + ffn(x1), *ln2)


def user_encoder(sd, input_embs, log_mask, cfg):
    """model/encoders.py:24-29 + modules.py:101-113."""
    b, t = log_mask.shape
    valid = (log_mask != 0)[:, None, None, :].expand(b, 1, t, t)
    allowed = torch.tril(valid)
    add_mask = torch.where(allowed, torch.tensor(0.0), torch.tensor(-1e9))
    x = layer_norm(input_embs + sd[UE + 'position_embedding.weight'][:t],
                   sd[UE + 'layer_norm.weight'], sd[UE + 'layer_norm.bias'], 1e-6)
    x = _drop(cfg, 'rows', 4000, x, 'p_sas')                # modules.py:104 dropout(layer_norm(input + position))
    i = 0
    if UE + 'transformer_blocks.com_dense2.weight' in sd:      # K-Adapter, model/model.py:562-583 SASRecKAdaptedTransformerBlocks
        last = 0
        while UE + f'transformer_blocks.adapter_list.{i}.down_project.weight' in sd:
            last = kadapter_block(sd, UE + f'transformer_blocks.adapter_list.{i}.', x + last, cfg['num_adapter_heads_sasrec'], cfg)
            x = sasrec_block(sd, UE + f'transformer_blocks.transformer_blocks.{i}.', x, add_mask, cfg)
            i += 1
        return linear(torch.cat([x, last], -1), sd[UE + 'transformer_blocks.com_dense2.weight'], sd[UE + 'transformer_blocks.com_dense2.bias'])
    while True:
        bp = UE + f'transformer_blocks.{i}.'
        if not any(k.startswith(bp) for k in sd):
            break
        x = sasrec_block(sd, bp, x, add_mask, cfg, site=4096 + 16 * i)
        i += 1
    return x


# --------------------------------------------------------------------------- training objective
def softplus(x):
    return torch.clamp(x, min=0) + torch.log1p(torch.exp(-x.abs()))


def score_loss(prec, tgt_pos, tgt_neg, log_mask, cfg):
    """model/model.py:62-68 (SASRec) / :127-133 (CPC): BCE-with-logits, mean over valid positions."""
    if cfg['arch'] == 'cpc':
        pos = (prec[:, -1] * tgt_pos[:, -1]).sum(-1)
        neg = (prec[:, -1] * tgt_neg[:, -1]).sum(-1)
        return softplus(-pos).mean() + softplus(neg).mean(), pos, neg
    pos = (prec * tgt_pos).sum(-1)
    neg = (prec * tgt_neg).sum(-1)
    idx = log_mask != 0
    return softplus(-pos[idx]).mean() + softplus(neg[idx]).mean(), pos, neg


def model_forward(sd, sample_items, log_mask, cfg):
    """model/model.py:48-70 Model.forward / :113-135 ModelCPC.forward."""
    e = cfg['embedding_dim']
    embs_all = image_encoder(sd, sample_items, cfg, noise=cfg.get('noise')) if cfg.get('tower', 'text') == 'image' \
        else text_encoder(sd, sample_items, cfg)
    embs = embs_all.view(-1, cfg['max_seq_len'] + 1, 2, e)
    pos_e, neg_e = embs[:, :, 0], embs[:, :, 1]
    prec = user_encoder(sd, pos_e[:, :-1], log_mask, cfg)
    loss, pos, neg = score_loss(prec, pos_e[:, 1:], neg_e[:, :-1], log_mask, cfg)
    return dict(loss=loss, pos_score=pos, neg_score=neg, prec_vec=prec, input_embs_all=embs_all)


def loss_and_grads(sd, trainable, sample_items, log_mask, cfg):
    """Forward + autograd backward of the restatement; returns (out, {name: grad})."""
    work = {k: v.detach().clone() for k, v in sd.items()}
    for k in trainable:
        work[k].requires_grad_(True)
    out = model_forward(work, sample_items, log_mask, cfg)
    grads = torch.autograd.grad(out['loss'], [work[k] for k in trainable], allow_unused=True)
    return out, {k: (g if g is not None else torch.zeros_like(work[k])) for k, g in zip(trainable, grads)}


# --------------------------------------------------------------------------- image tower
# HF ViTForImageClassification / ViTMAEModel are third party (transformers==4.20.1, README.md:64); the reference reaches
# them at Downstream/CV/model/encoders.py:21-32 and wraps their sub-modules at Downstream/CV/model/model.py:182-212,
# 432-462.  The restatement follows 4.20.1's pre-LN layer algebra; it is pinned (tools/gen_golden_cv.py) against the
# installed HF ViT / ViT-MAE forward for the un-adapted backbone and against the reference's own wrapper classes, Model
# and User_Encoder for everything around it.
def _vit_prefix(sd):
    return 'cv_encoder.image_net.vit.' if 'cv_encoder.image_net.vit.layernorm.weight' in sd else 'cv_encoder.image_net.'


def vit_sub_output(sd, p, h, cfg, inp=None):
    """(possibly wrapped) ViTSelfOutput / ViTOutput at prefix p: dense [-> adapter]; the residual is added by the caller
    (ViTLayer for attention.output, the wrapper / ViTOutput itself for output -- same sum either way)."""
    if p + 'self_output.dense.weight' in sd:
        h = linear(h, sd[p + 'self_output.dense.weight'], sd[p + 'self_output.dense.bias'])
        a = p + 'adapter.'
        if cfg.get('is_serial', 'True') == 'None' and cfg.get('adapter_type', 'houslby') == 'houslby':
            return h + houlsby_block(sd, a, inp, cfg)      # model.py:165-179 VITAdaptedParallelOutput: + adapter(input_tensor)
        if a + 'down_sampler.W_left' in sd:          # model.py:432-462 (HyperComplexAdapterBlock: no inner residual)
            return compacter_block(sd, a, h, cfg)
        return houlsby_block(sd, a, h, cfg)          # model.py:182-212
    return linear(h, sd[p + 'dense.weight'], sd[p + 'dense.bias'])


def vit_embed(sd, images, cfg, noise=None):
    """HF ViTEmbeddings / ViTMAEEmbeddings (4.20.1).  MAE: patches + pos[1:], keep the int(N (1 - ratio)) patches of
    smallest noise in argsort order, prepend cls + pos[0]."""
    p = _vit_prefix(sd) + 'embeddings.'
    if p + 'Prompt_Tokens' in sd:                     # Downstream/CV/model/model.py:512-535 SoftPrompt: tokens appended after cls + patches
        base = vit_embed({k.replace('embeddings.wte.', 'embeddings.'): v for k, v in sd.items() if 'Prompt_Tokens' not in k and '.embeddings.patch_embeddings.' not in k},
                         images, cfg, noise)
        return torch.cat([base, sd[p + 'Prompt_Tokens'].expand(base.shape[0], -1, -1)], 1)
    w, b = sd[p + 'patch_embeddings.projection.weight'], sd[p + 'patch_embeddings.projection.bias']
    x = torch.nn.functional.conv2d(images, w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2)
    n = x.shape[0]
    pos, cls = sd[p + 'position_embeddings'], sd[p + 'cls_token']
    if cfg.get('mae'):
        x = x + pos[:, 1:]
        keep = torch.argsort(noise, dim=1)[:, :int(x.shape[1] * (1 - cfg.get('mask_ratio', 0.75)))]
        x = torch.gather(x, 1, keep[:, :, None].expand(-1, -1, x.shape[2]))
        return torch.cat([(cls + pos[:, :1]).expand(n, -1, -1), x], 1)
    return torch.cat([cls.expand(n, -1, -1), x], 1) + pos


def vit_layer(sd, lp, x, cfg):
    """HF ViTLayer (pre-LN): x + attn(LN(x)); then + mlp(LN(.)).  q / v may be loralib Linears (run_adapter.py:384-388)."""
    eps, nh = cfg.get('vit_ln_eps', 1e-12), cfg['vit_heads']
    n, s, hdim = x.shape
    n1 = layer_norm(x, sd[lp + 'layernorm_before.weight'], sd[lp + 'layernorm_before.bias'], eps)
    a = lp + 'attention.attention.'
    r = cfg.get('lora_r_vit', 0)
    q, k, v = lora_linear(sd, a + 'query.', n1, r), linear(n1, sd[a + 'key.weight'], sd[a + 'key.bias']), lora_linear(sd, a + 'value.', n1, r)
    sp = lambda t: t.view(n, s, nh, hdim // nh).transpose(1, 2)
    pr = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(hdim // nh), -1)
    ctx = (pr @ sp(v)).transpose(1, 2).reshape(n, s, hdim)
    x1 = vit_sub_output(sd, lp + 'attention.output.', ctx, cfg) + x
    n2 = layer_norm(x1, sd[lp + 'layernorm_after.weight'], sd[lp + 'layernorm_after.bias'], eps)
    u = gelu_erf(linear(n2, sd[lp + 'intermediate.dense.weight'], sd[lp + 'intermediate.dense.bias']))
    return vit_sub_output(sd, lp + 'output.', u, cfg, inp=x1) + x1


def vit_encode(sd, images, cfg, noise=None, return_all=False):
    p = _vit_prefix(sd)
    kad = p + 'encoder.com_dense.weight' in sd      # Downstream/CV/model/model.py:374-404 VITKAdaptedCVModel sits where vit.encoder was
    ep = p + ('encoder.vit_encoder.' if kad else 'encoder.')
    x = vit_embed(sd, images, cfg, noise)
    hs = [x]                                        # 4.20.1 ViTEncoder: hidden_states[0] = embedding output, [i + 1] = output of layer i
    i = 0
    while f'{ep}layer.{i}.layernorm_before.weight' in sd:
        x = vit_layer(sd, f'{ep}layer.{i}.', x, cfg)
        hs.append(x)
        i += 1
    if kad:                                         # :393-401: adapters chained over the listed hidden states, then com_dense([last ; adapter])
        last = 0
        for j, k in enumerate(int(t) + 1 for t in str(cfg['k_adapter_bert_list']).split(',')):
            last = kadapter_block(sd, p + f'encoder.bert_adapter_list.{j}.', hs[k] + last, cfg['num_adapter_heads_bert'], cfg)
        x = linear(torch.cat([x, last], -1), sd[p + 'encoder.com_dense.weight'], sd[p + 'encoder.com_dense.bias'])
    x = layer_norm(x, sd[p + 'layernorm.weight'], sd[p + 'layernorm.bias'], cfg.get('vit_ln_eps', 1e-12))
    return (x, hs) if return_all else x


def image_encoder(sd, images, cfg, noise=None):
    """Downstream/CV/model/encoders.py:21-22 (MAE_Encoder: GELU(cv_proj(last_hidden[:, 0]))) / :31-32 (Vit_Encoder:
    GELU(classifier(layernorm(x)[:, 0])))."""
    cls = vit_encode(sd, images, cfg, noise)[:, 0]
    if cfg.get('mae'):
        return gelu_erf(linear(cls, sd['cv_encoder.cv_proj.weight'], sd['cv_encoder.cv_proj.bias']))
    return gelu_erf(linear(cls, sd['cv_encoder.image_net.classifier.weight'], sd['cv_encoder.image_net.classifier.bias']))


def normalize_u8(img_u8_hwc):
    """Downstream/CV/data_utils/dataset.py:77-81 without the Resize (source already R x R): ToTensor + Normalize(0.5, 0.5)."""
    return ((img_u8_hwc.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2).contiguous()


def lr_group_cv(name, cfg_lrs):
    """Downstream/CV/run_adapter.py:491-517."""
    ad = 'adapter' in name
    if 'image_net' in name and not ('fc' in name or 'classifier' in name or 'decoder_pred' in name):
        return cfg_lrs['adapter_cv_lr'] if ad else cfg_lrs['fine_tune_lr']
    return cfg_lrs['adapter_sasrec_lr'] if ad else cfg_lrs['lr']


# --------------------------------------------------------------------------- optimiser
def lr_group(name, cfg_lrs):
    """Downstream/Text/run.py:510-529 grouping by substrings of the parameter name."""
    if 'bert_encoder' in name:
        return cfg_lrs['adapter_bert_lr'] if ('adapter' in name or 'lora' in name) else cfg_lrs['fine_tune_lr']
    return cfg_lrs['adapter_sasrec_lr'] if ('adapter' in name or 'lora' in name) else cfg_lrs['lr']


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) as called at run.py:524-529,600."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps))


def train_steps(sd, trainable, batches, cfg, lrs, n_steps, group=None):
    """n Adam steps on the same restatement; returns the list of losses and the final trainable tensors."""
    sd = {k: v.detach().clone() for k, v in sd.items()}
    ms = {k: torch.zeros_like(sd[k]) for k in trainable}
    vs = {k: torch.zeros_like(sd[k]) for k in trainable}
    losses = []
    for s in range(1, n_steps + 1):
        items, mask = batches[(s - 1) % len(batches)]
        out, grads = loss_and_grads(sd, trainable, items, mask, cfg)
        losses.append(float(out['loss'].detach()))
        for k in trainable:
            adam_step(sd[k], grads[k], ms[k], vs[k], s, (group or lr_group)(k, lrs))
    return losses, {k: sd[k] for k in trainable}


# --------------------------------------------------------------------------- data (a1) and eval (a13)
def build_train_sample(seq, item_num, max_seq_len, rng=random):
    """data_utils/dataset.py:24-49 BuildTrainDataset.__getitem__ (ids only; caller gathers item_content)."""
    L = max_seq_len + 1
    pad = L - len(seq)
    n_tok = len(seq) - 1
    log_mask = [0] * pad + [1] * n_tok
    negs = []
    for _ in range(n_tok):
        s = rng.randint(1, item_num)
        while s in seq:
            s = rng.randint(1, item_num)
        negs.append(s)
    ids = np.array([[0] * pad + list(seq), [0] * pad + negs + [0]]).T      # [L, 2]
    return ids, np.array(log_mask, dtype=np.float32)


def split_sequences(user_seq, max_seq_len):
    """data_utils/preprocess.py:48-59."""
    train = user_seq[:-2]
    valid = user_seq[-(max_seq_len + 2):-1]
    test = user_seq[-(max_seq_len + 1):]
    return train, valid, test, train, user_seq[:-1]


def item_embeddings(sd, item_content, cfg, bs=512):
    """data_utils/metrics.py:62-79."""
    out = []
    with torch.no_grad():
        for i in range(0, item_content.shape[0], bs):
            out.append(text_encoder(sd, torch.as_tensor(item_content[i:i + bs]).long(), cfg))
    return torch.cat(out, 0)


def eval_ranks(sd, item_emb, eval_seq, history, cfg):
    """data_utils/metrics.py:82-116 + dataset.py:52-78: rank of the held-out target among all items.

    rank = 1 + #{items != history, != pad column 0, scoring strictly above the target}
    (the reference derives it through argsort of the score vector; ties are not broken stably there).
    """
    L = cfg['max_seq_len'] + 1
    users = sorted(eval_seq)
    ranks = []
    with torch.no_grad():
        for u in users:
            seq = list(eval_seq[u])
            toks, target = seq[:-1], seq[-1]
            pad = L - len(seq)
            ids = [0] * pad + toks
            mask = torch.tensor([[0.0] * pad + [1.0] * len(toks)])
            prec = user_encoder(sd, item_emb[ids][None], mask, cfg)[0, -1]
            score = item_emb @ prec
            score[torch.as_tensor(history[u]).long()] = -float('inf')
            score = score[1:]
            ranks.append(int((score > score[target - 1]).sum()) + 1)
    return users, np.array(ranks)


def hit_ndcg(ranks, topk=10):
    """data_utils/metrics.py:51-59."""
    ranks = np.asarray(ranks)
    hit = (ranks <= topk).astype(np.float64)
    ndcg = np.where(ranks <= topk, 1.0 / np.log2(ranks + 1.0), 0.0)
    return float(hit.mean()), float(ndcg.mean())
