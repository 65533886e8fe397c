#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference (CPU, build container only).

Runs only where /root/reference exists; nothing here travels to the GPU box except the
.npz files it writes (inputs + expected outputs = data, no reference source).

  python tools/gen_golden.py            # writes tests/golden/{base,<variant>,dataset,eval}.npz

Follows SURVEY.md Appendix B: the harness re-creates the ~30 lines of adapter injection
(Downstream/Text/run.py:385-479) and optimiser grouping (run.py:505-529) because run.py
itself needs CUDA + NCCL at import-time globals.
"""
import argparse
import copy
import logging
import os
import random
import sys

import numpy as np
import torch

REF = '/root/reference/Downstream/Text'
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

from transformers import BertConfig, BertModel, RobertaConfig, RobertaModel  # noqa: E402
import model as refm  # noqa: E402
from model import (Model, ModelCPC, BertAdaptedSelfOutput, SASRecAdaptedSelfOutput,  # noqa: E402
                   BertAdaptedParallelSelfOutput, SASRecParallelAdaptedSelfOutput,
                   BertPfeifferAdaptedSelfOutput, SASRecPfeifferAdaptedSelfOutput,
                   SASRecPfeifferVer2AdaptedSelfOutput,
                   BertCompacterAdaptedSelfOutput, SASRecCompacterAdaptedSelfOutput, PHMLinear)
import data_utils  # noqa: E402
from data_utils import BuildTrainDataset, eval_model, get_item_embeddings  # noqa: E402
import data_utils.metrics as ref_metrics  # noqa: E402

VOCAB, HID, LAYERS, HEADS, FFN, MAXPOS = 120, 128, 2, 2, 256, 40
ITEM_NUM, B, L, NW = 200, 4, 21, 30
LRS = dict(fine_tune_lr=5e-5, lr=1e-4, adapter_bert_lr=1.5e-4, adapter_sasrec_lr=1.5e-4)


def make_args(**kw):
    a = argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1,
        transformer_block=2, num_words_title=NW, num_words_abstract=50, num_words_body=50,
        news_attributes=['title'], word_embedding_dim=HID, bert_model_load='bert_tiny',
        bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation='RELU', hypercomplex_division=4, phm_init_range=1e-4,
        adapter_type='houslby', is_serial='True', num_workers=0, arch='sasrec')
    for k, v in kw.items():
        setattr(a, k, v)
    return a


class CompacterModel(torch.nn.Module):  # what Downstream/Text/run.py:70-83 does
    def __init__(self, args, model):
        super().__init__()
        n = args.hypercomplex_division
        self.model = model
        self.phm_rule = torch.nn.Parameter(torch.FloatTensor(n, n, n), requires_grad=True)
        self.phm_rule.data.normal_(mean=0, std=args.phm_init_range)
        for _, sub in model.named_modules():
            if isinstance(sub, PHMLinear):
                sub.set_phm_rule(phm_rule=self.phm_rule)

    def forward(self, sample_items, log_mask, local_rank):
        return self.model(sample_items, log_mask, local_rank)


def inject(model, args):
    """Downstream/Text/run.py:385-479, CPU, without .to(local_rank)."""
    layers = model.bert_encoder.text_encoders.title.bert_model.encoder.layer
    blocks = model.user_encoder.transformer_encoder.transformer_blocks
    t = args.adapter_type
    if t == 'none':
        return model
    if 'kadapter' in t:                     # run.py:409-413
        te = model.bert_encoder.text_encoders.title
        te.bert_model = refm.model.BertKAdaptedBertModel(te.bert_model, args)
        ue = model.user_encoder.transformer_encoder
        ue.transformer_blocks = refm.model.SASRecKAdaptedTransformerBlocks(ue.transformer_blocks, args)
        return model
    if 'prompt' in t:                       # run.py:429-434
        bm = model.bert_encoder.text_encoders.title.bert_model
        bm.set_input_embeddings(refm.model.SoftEmbedding(bm.get_input_embeddings(), n_tokens=args.n_tokens, initialize_from_vocab=True))
        return model
    if 'pfeiffer_ver2' in t:
        for lyr in layers:
            lyr.attention.output = BertAdaptedSelfOutput(lyr.attention.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferVer2AdaptedSelfOutput(blk, args)
    elif 'pfeiffer' in t:
        for lyr in layers:
            lyr.output = BertPfeifferAdaptedSelfOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecPfeifferAdaptedSelfOutput(blk, args)
    elif 'compacter' in t:
        for lyr in layers:
            lyr.attention.output = BertCompacterAdaptedSelfOutput(lyr.attention.output, args)
            lyr.output = BertCompacterAdaptedSelfOutput(lyr.output, args)
        for i, blk in enumerate(blocks):
            blocks[i] = SASRecCompacterAdaptedSelfOutput(blk, args)
        model = CompacterModel(args, model)
    elif 'houslby' in t:
        if 'None' not in args.is_serial:
            for lyr in layers:
                lyr.attention.output = BertAdaptedSelfOutput(lyr.attention.output, args)
                lyr.output = BertAdaptedSelfOutput(lyr.output, args)
            for i, blk in enumerate(blocks):
                blocks[i] = SASRecAdaptedSelfOutput(blk, args)
        else:
            for lyr in layers:
                lyr.attention.output = BertAdaptedParallelSelfOutput(lyr.attention.output, args)
                lyr.output = BertAdaptedParallelSelfOutput(lyr.output, args)
            for i, blk in enumerate(blocks):
                blocks[i] = SASRecParallelAdaptedSelfOutput(blk, args)
    return model


def base_name(k):
    """adapted state_dict key -> key of the un-adapted model that holds the same tensor."""
    if k.startswith('model.'):
        k = k[len('model.'):]
    return k.replace('.self_output.', '.').replace('.transformer_block.', '.').replace('.word_embeddings.wte.', '.word_embeddings.') \
        .replace('.bert_model.bert_model.', '.bert_model.').replace('.transformer_blocks.transformer_blocks.', '.transformer_blocks.')


def make_content(rng, n_items, roberta=False):
    """item_content [n_items+1, 2*NW]: ids || mask; item 0 = PAD item (all zeros)."""
    c = np.zeros((n_items + 1, 2 * NW), dtype=np.int64)
    for i in range(1, n_items + 1):
        n = int(rng.integers(4, NW + 1))
        ids = rng.integers(5, VOCAB, size=n)
        if roberta:
            ids[0], ids[-1] = 0, 2
            c[i, :NW] = 1                      # RoBERTa pad id
        else:
            ids[0], ids[-1] = 3, 4             # [CLS]/[SEP] stand-ins
        c[i, :n] = ids
        c[i, NW:NW + n] = 1
    if roberta:
        c[0, :NW] = 0                          # reference pads item 0 with zeros regardless
    return c


def make_batch(rng, content, seed):
    random.seed(seed)
    u2seq = {}
    lens = [21, 21, 9, 5]
    for u in range(B):
        u2seq[u] = [int(x) for x in rng.choice(np.arange(1, ITEM_NUM + 1), size=lens[u], replace=False)]
    ds = BuildTrainDataset(u2seq=u2seq, item_content=content, item_num=ITEM_NUM, max_seq_len=20, use_modal=True)
    random.seed(seed)
    items, masks = zip(*[ds[u] for u in range(B)])
    return u2seq, torch.stack(items), torch.stack(masks)


def optimizer_for(model):
    """run.py:505-529 (names carry the DDP 'module.' prefix there; substrings are unaffected)."""
    g = {k: [] for k in ('bert', 'rec', 'abert', 'arec')}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ad = 'adapter' in name or 'lora' in name
        if 'bert_encoder' in name:
            g['abert' if ad else 'bert'].append(p)
        else:
            g['arec' if ad else 'rec'].append(p)
    return torch.optim.Adam([
        {'params': g['bert'], 'lr': LRS['fine_tune_lr']}, {'params': g['rec'], 'lr': LRS['lr']},
        {'params': g['abert'], 'lr': LRS['adapter_bert_lr']}, {'params': g['arec'], 'lr': LRS['adapter_sasrec_lr']}])


def run_variant(name, base_model, content, items, masks, args, hidden_dump=False):
    torch.manual_seed(1000 + sum(map(ord, name)))
    m = copy.deepcopy(base_model)
    if args.arch == 'cpc':
        cpc = ModelCPC(args, ITEM_NUM, True, m.bert_encoder.text_encoders.title.bert_model)
        cpc.bert_encoder = m.bert_encoder
        cpc.user_encoder = m.user_encoder
        m = cpc
    for p in m.parameters():
        p.requires_grad = False
    m = inject(m, args)
    if args.adapter_type == 'none':
        for p in m.parameters():
            p.requires_grad = True
        for n_, p in m.named_parameters():
            if 'pooler' in n_:
                p.requires_grad = False
    # adapters' default init (N(0,1e-2) / zeros bias) makes bias grads the only large ones; jitter so that
    # every trainable tensor has a non-trivial value and gradient.
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.requires_grad and ('adapter' in n_ or n_.endswith('phm_rule') or n_.startswith('LN') or '.LN.' in n_ or 'learned_embedding' in n_ or 'com_dense' in n_):
                p.add_(0.05 * torch.randn_like(p))
    m.eval()
    inner = m.model if isinstance(m, CompacterModel) else m
    sample = items.view(-1, 2 * NW)
    out = {}
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    base_sd = base_model.state_dict()
    new = {}
    for k, v in sd.items():
        bk = base_name(k)
        if bk in base_sd and torch.equal(base_sd[bk], v):
            continue
        new[k] = v.numpy()
    out['all_keys'] = np.array(list(sd.keys()))
    for k, v in new.items():
        out['sd/' + k] = v
    trainable = [n_ for n_, p in m.named_parameters() if p.requires_grad]
    out['trainable'] = np.array(trainable)

    # forward pieces
    with torch.no_grad():
        embs = inner.bert_encoder(sample)
        e = embs.view(-1, L, 2, 64)
        prec = inner.user_encoder(e[:, :-1, 0], masks, 'cpu')
    out['input_embs_all'] = embs.numpy()
    out['prec_vec'] = prec.numpy()
    if hidden_dump:
        bm = inner.bert_encoder.text_encoders.title.bert_model
        with torch.no_grad():
            hs = bm(input_ids=sample[:8, :NW], attention_mask=sample[:8, NW:], output_hidden_states=True).hidden_states
        out['hidden_states'] = torch.stack(hs).numpy()
    m.zero_grad()
    loss = m(sample, masks, 'cpu')
    loss.backward()
    out['loss'] = loss.detach().numpy()
    if args.adapter_type != 'none':
        for n_, p in m.named_parameters():
            if p.requires_grad:
                out['grad/' + n_] = p.grad.detach().numpy().copy()
    else:   # full fine-tune: keep a handful of grads only (fixtures stay small)
        keep = ['fc.weight', 'layer.1.output.dense.weight', 'layer.0.attention.self.query.weight',
                'position_embedding.weight', 'blocks.0.multi_head_attention.w_Q.weight', 'embeddings.LayerNorm.weight']
        for n_, p in m.named_parameters():
            if p.requires_grad and any(n_.endswith(s) for s in keep):
                out['grad/' + n_] = p.grad.detach().numpy().copy()
    # 3 Adam steps on the same batch (eval mode = dropout off), 4 lr groups
    if args.adapter_type != 'none':
        state0 = copy.deepcopy(m.state_dict())
        opt = optimizer_for(m)
        losses = []
        for s in range(3):
            opt.zero_grad()
            l_ = m(sample, masks, 'cpu')
            l_.backward()
            opt.step()
            losses.append(float(l_.detach()))
            if s in (0, 2):
                for n_, p in m.named_parameters():
                    if p.requires_grad:
                        out[f'adam{s + 1}/' + n_] = p.detach().numpy().copy()
        out['adam_losses'] = np.array(losses)
        m.load_state_dict(state0)          # the returned model keeps the pre-Adam weights (used by gen_eval)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: loss {float(loss):.6f}  trainable {len(trainable)} tensors  new keys {len(new)}')
    return m


def gen_eval(model, args, content, rng):
    """F7: reference get_item_embeddings + eval_model under a 1-process gloo group."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('gloo', rank=0, world_size=1)

    class Wrap:  # the functions only dereference `.module`
        def __init__(self, m):
            self.module = m

        def eval(self):
            self.module.eval()

    n_users = 50
    full = {}
    for u in range(n_users):
        n = int(rng.integers(5, 24))
        full[u] = [int(x) for x in rng.choice(np.arange(1, ITEM_NUM + 1), size=n, replace=False)]
    users_valid, users_test, hist_valid, hist_test = {}, {}, {}, {}
    for u, seq in full.items():   # data_utils/preprocess.py:48-59
        users_valid[u] = seq[-22:-1]
        users_test[u] = seq[-21:]
        hist_valid[u] = torch.LongTensor(np.array(seq[:-2]))
        hist_test[u] = torch.LongTensor(np.array(seq[:-1]))
    log = logging.getLogger('golden')
    w = Wrap(model)
    emb = get_item_embeddings(w, content, 64, args, True, 'cpu')
    res = {}
    orig = ref_metrics.metrics_topK
    for tag, seqs, hist in (('valid', users_valid, hist_valid), ('test', users_test, hist_test)):
        rec = []

        def spy(y_score, y_true, item_rank, topK, local_rank):
            r = orig(y_score, y_true, item_rank, topK, local_rank)
            rec.append(r.numpy().copy())
            return r
        ref_metrics.metrics_topK = spy
        means = []
        orig_print = ref_metrics.print_metrics
        ref_metrics.print_metrics = lambda x, lf, vt: means.append(list(x))
        hr = eval_model(w, hist, seqs, emb, 16, args, ITEM_NUM, log, tag, 'cpu')
        ref_metrics.print_metrics = orig_print
        ref_metrics.metrics_topK = orig
        res[tag + '_hit_ndcg_per_user'] = np.array(rec)
        res[tag + '_hr10'] = np.array(hr)
        res[tag + '_means'] = np.array(means[0])
    res['item_embeddings'] = emb.numpy()
    res['full_seq_flat'] = np.array([x for u in range(n_users) for x in full[u]])
    res['full_seq_len'] = np.array([len(full[u]) for u in range(n_users)])
    np.savez_compressed(os.path.join(OUT, 'eval.npz'), **res)
    print('eval: valid', res['valid_means'], 'test', res['test_means'])
    dist.destroy_process_group()


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(123456)
    torch.manual_seed(123456)
    cfg = BertConfig(vocab_size=VOCAB, hidden_size=HID, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                     intermediate_size=FFN, max_position_embeddings=MAXPOS, attn_implementation='eager',
                     output_hidden_states=(len(sys.argv) > 2 and sys.argv[2] == 'kadapter'))
    args = make_args()
    base = Model(args, ITEM_NUM, True, BertModel(cfg))
    if len(sys.argv) > 2 and sys.argv[1] == '--only' and sys.argv[2] == 'roberta_prompt':
        # soft prompt on RoBERTa: the position ids still come from the ORIGINAL token ids (titles shorter than n_tokens have pads inside the prompt)
        rcfg = RobertaConfig(vocab_size=VOCAB, hidden_size=HID, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                             intermediate_size=FFN, max_position_embeddings=MAXPOS + 2, type_vocab_size=1,
                             layer_norm_eps=1e-5, pad_token_id=1, attn_implementation='eager')
        rargs = make_args(adapter_type='prompt', n_tokens=8, arch='cpc', bert_model_load='roberta_tiny')
        rbase = Model(rargs, ITEM_NUM, True, RobertaModel(rcfg))
        fx = np.load(os.path.join(OUT, 'base_roberta.npz'))
        rbase.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('sd/')})
        rbase.eval()
        run_variant('roberta_prompt', rbase, fx['item_content'], torch.from_numpy(fx['sample_items']), torch.from_numpy(fx['log_mask']), rargs)
        return
    if len(sys.argv) > 2 and sys.argv[1] == '--only':        # later additions: rebuild the base from base.npz, write ONE new fixture
        fx = np.load(os.path.join(OUT, 'base.npz'))
        base.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('sd/')})
        base.eval()
        items, masks = torch.from_numpy(fx['sample_items']), torch.from_numpy(fx['log_mask'])
        if sys.argv[2] == 'prompt':
            run_variant('prompt', base, fx['item_content'], items, masks, make_args(adapter_type='prompt', n_tokens=8))
        if sys.argv[2] == 'kadapter':
            run_variant('kadapter', base, fx['item_content'], items, masks,
                        make_args(adapter_type='kadapter', k_adapter_bert_list='0,1', k_adapter_bert_hidden_dim=64, num_adapter_heads_bert=4,
                                  num_adapter_heads_sasrec=2))
        return
    # HF inits LayerNorm to (1, 0) and biases to 0; jitter so that those terms are exercised
    with torch.no_grad():
        for n_, p in base.named_parameters():
            if 'LayerNorm' in n_ or 'layer_norm' in n_ or n_.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    base.eval()
    content = make_content(rng, ITEM_NUM)
    u2seq, items, masks = make_batch(rng, content, seed=7)
    np.savez_compressed(
        os.path.join(OUT, 'base.npz'),
        **{'sd/' + k: v.numpy() for k, v in base.state_dict().items()},
        item_content=content, sample_items=items.numpy(), log_mask=masks.numpy(),
        geometry=np.array([VOCAB, HID, LAYERS, HEADS, FFN, MAXPOS, ITEM_NUM, B]))
    # F1 dataset fixture: (u2seq, seed) -> (item ids, log_mask)
    np.savez_compressed(
        os.path.join(OUT, 'dataset.npz'), seed=np.array(7),
        seq_flat=np.array([x for u in range(B) for x in u2seq[u]]), seq_len=np.array([len(u2seq[u]) for u in range(B)]),
        item_content=content, sample_items=items.numpy(), log_mask=masks.numpy())

    variants = [
        ('houlsby', make_args(), True),
        ('houlsby_gelu', make_args(adapter_activation='GELU'), False),
        ('houlsby_parallel', make_args(is_serial='None'), False),
        ('pfeiffer', make_args(adapter_type='pfeiffer', adapter_activation='relu'), False),
        ('pfeiffer_ver2', make_args(adapter_type='pfeiffer_ver2'), False),
        ('compacter', make_args(adapter_type='compacter'), False),
        ('houlsby_cpc', make_args(arch='cpc'), False),
        ('finetune_all', make_args(adapter_type='none'), False),
    ]
    models = {}
    for name, a, hd in variants:
        models[name] = run_variant(name, base, content, items, masks, a, hidden_dump=hd)

    # RoBERTa + CPC + Pfeiffer (BASELINE config 4), separate base (pad id 1, 1 token type, eps 1e-5)
    torch.manual_seed(4242)
    rcfg = RobertaConfig(vocab_size=VOCAB, hidden_size=HID, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                         intermediate_size=FFN, max_position_embeddings=MAXPOS + 2, type_vocab_size=1,
                         layer_norm_eps=1e-5, pad_token_id=1, attn_implementation='eager')
    rargs = make_args(adapter_type='pfeiffer', adapter_activation='relu', arch='cpc', bert_model_load='roberta_tiny')
    rbase = Model(rargs, ITEM_NUM, True, RobertaModel(rcfg))
    with torch.no_grad():
        for n_, p in rbase.named_parameters():
            if 'LayerNorm' in n_ or 'layer_norm' in n_ or n_.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    rbase.eval()
    rcontent = make_content(rng, ITEM_NUM, roberta=True)
    _, ritems, rmasks = make_batch(rng, rcontent, seed=11)
    np.savez_compressed(
        os.path.join(OUT, 'base_roberta.npz'),
        **{'sd/' + k: v.numpy() for k, v in rbase.state_dict().items()},
        item_content=rcontent, sample_items=ritems.numpy(), log_mask=rmasks.numpy())
    run_variant('roberta_cpc_pfeiffer', rbase, rcontent, ritems, rmasks, rargs)

    gen_eval(models['houlsby'], make_args(), content, rng)


if __name__ == '__main__':
    main()
