"""Parity in the mode and at the size that is benchmarked:

  * one training step at BERT-base geometry (12 x H=768, 12 heads, F=3072, S=30; B = 2 users = 84 items): the fp32 instantiation
    of the HIP path vs the CPU oracle at the north_star tolerance (1e-4), and the bf16 instantiation (what bench.py times) vs both,
    with the MEASURED bf16 bound asserted and printed (DESIGN.md section 2 records it);
  * evaluation on 2 000 items x 600 users with weights conditioned so that HR@10 is far from 0 (reference metric:
    Downstream/Text/data_utils/metrics.py:82-116): HR@10 / nDCG@10 and per-user ranks, fp32 HIP and bf16 HIP vs the fp32 oracle.
"""
import argparse
import logging

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def base_args(dtype, act='RELU'):
    return argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=768,
        bert_model_load='bert_base_uncased', bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation=act, hypercomplex_division=4, phm_init_range=1e-4, adapter_type='houslby', is_serial='True',
        adding_adapter_to='all', arch='sasrec', compute_dtype=dtype)


def build_base(seed=3, users=2, n_items=4096, act='RELU'):
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    torch.manual_seed(seed)
    model = Model(base_args('fp32', act), n_items, True, BertBackbone(BERT_BASE))
    freeze_all(model)
    model = inject_adapters(model, model.args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:                       # adapter biases / fc_up start at 0 / 1e-2: give every gradient path a signal
                p.add_(0.02 * torch.randn_like(p))
    model.eval()
    g = torch.Generator().manual_seed(seed)
    L = 21
    ids = torch.zeros(users, L, 2, 60, dtype=torch.int64)
    mask = torch.zeros(users, L - 1)
    for u in range(users):
        n = L if u == 0 else 9                        # one full history, one short (left-padded with the PAD item)
        for slot in range(L - n, L):
            for side in range(2):
                if side == 1 and slot == L - 1:
                    continue
                ln = 30 if (slot + side) % 3 else int(torch.randint(4, 30, (1,), generator=g))      # full and partially padded titles
                ids[u, slot, side, 0] = 101
                ids[u, slot, side, 1:ln - 1] = torch.randint(1000, 30000, (ln - 2,), generator=g)
                ids[u, slot, side, ln - 1] = 102
                ids[u, slot, side, 30:30 + ln] = 1
        mask[u, L - n:] = 1
    return model, ids.view(-1, 60), mask


def hip_step(model, dtype, items, mask):
    model.compute_dtype = dtype
    model.invalidate_native()
    for p in model.parameters():
        p.grad = None
    model.to(DEV)
    model.eval()
    loss = model(items.to(DEV), mask.to(DEV), DEV)
    pos, neg = model._engine().scores()
    loss.backward()
    emb = model.bert_encoder(items.to(DEV)).cpu()
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    out = dict(loss=float(loss.detach()), pos=pos.cpu(), neg=neg.cpu(), emb=emb, grads=grads)
    model.cpu()
    return out


def grad_err(a, b):
    """worst over tensors of max|a - b| / max|b|, and the tensor it occurs in."""
    worst, where = 0.0, ''
    for n, r in b.items():
        e = float((a[n] - r).abs().max() / r.abs().max().clamp_min(1e-30))
        if e > worst:
            worst, where = e, n
    return worst, where


@pytest.mark.parametrize('act', ['GELU', 'RELU'])
def test_bert_base_geometry_step_fp32_and_bf16_vs_oracle(act):
    """act = RELU is the reference's default (parameters.py:64) and what bench.py runs; its derivative is discontinuous at 0, so two
    fp32 implementations that differ in summation order disagree on act'(zp) for the few pre-activations within rounding of 0: each
    such flip moves one token's contribution (1 / 2 520 of a row of dW_down here) -- the gradient bound for RELU is therefore
    ~1 / n_tokens, not 1e-4; with the smooth GELU adapter the same step meets 1e-4 everywhere."""
    from oracle import ref_cpu as R
    model, items, mask = build_base(act=act)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(names) == 24 * 4 + 4 * 4                           # 24 BERT adapters + 4 SASRec adapters, 4 tensors each
    out, grads = R.loss_and_grads(sd, names, items, mask, dict(R.DEFAULT_CFG, adapter_activation=act))
    ref = dict(loss=float(out['loss'].detach()), pos=out['pos_score'].detach(), neg=out['neg_score'].detach(),
               emb=out['input_embs_all'].detach(), grads=grads)
    valid = mask.bool()

    def diffs(a, b):
        g, where = grad_err(a['grads'], b['grads'])
        return dict(loss=abs(a['loss'] - b['loss']), pos=float((a['pos'][valid] - b['pos'][valid]).abs().max()),
                    neg=float((a['neg'][valid] - b['neg'][valid]).abs().max()), emb=float((a['emb'] - b['emb']).abs().max()),
                    grad=g, grad_where=where)
    f32 = hip_step(model, 'fp32', items, mask)
    d32 = diffs(f32, ref)
    print(f'BERT-base {act} fp32 HIP vs oracle:', d32)
    # north_star tolerance, fp32 instantiation of the same kernels (12 layers deep)
    assert d32['loss'] < 1e-4 and d32['pos'] < 1e-4 and d32['neg'] < 1e-4 and d32['emb'] < 1e-4, d32
    assert d32['grad'] < (1e-4 if act == 'GELU' else 2e-3), d32
    b16 = hip_step(model, 'bf16', items, mask)
    d16o, d16f = diffs(b16, ref), diffs(b16, f32)
    print(f'BERT-base {act} bf16 HIP vs oracle:', d16o)
    print(f'BERT-base {act} bf16 HIP vs fp32 HIP:', d16f)
    print(f"|score| scale: max |pos| {float(ref['pos'][valid].abs().max()):.3f}, loss {ref['loss']:.4f}, max |emb| {float(ref['emb'].abs().max()):.3f}")
    # bf16 storage / fp32 accumulate at full depth: the bound is the measured one with ~2x headroom.
    # ~10 roundings of 2^-9 per layer x 12 post-LN layers on O(1) activations.
    assert d16o['loss'] < 3e-2 and d16o['emb'] < 4e-2 and d16o['pos'] < 0.15 and d16o['neg'] < 0.15, d16o
    assert d16o['grad'] < 0.2, d16o
    assert abs(d16o['loss'] - d16f['loss']) < 1e-4                # the two fp32 references agree with each other


def build_eval_case(n_items=2000, n_users=600, seed=5):
    """tiny-geometry BERT + Houlsby with weights conditioned so that held-out targets are predictable: items come in near-twin
    pairs (titles differ in 3 of 28 tokens) and 70 % of the users' targets are the twin of their last history item; the word
    embeddings, the attention value/output weights and the item head are scaled so that item embeddings differ between items
    (a random-init encoder gives nearly identical embeddings for all items and HR@10 = 0)."""
    import test_engine_gpu as TG
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model
    args = TG.make_args(compute_dtype='fp32')
    torch.manual_seed(seed)
    model = Model(args, n_items, True, BertBackbone(dict(TG.GEOM, vocab_size=500)))
    freeze_all(model)
    model = inject_adapters(model, args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
            if n.endswith('attention.self.value.weight') or ('attention.output' in n and n.endswith('dense.weight')):
                p.mul_(8.0)
            if n.endswith('word_embeddings.weight'):
                p.mul_(30.0)
            if n.endswith('title.fc.weight'):
                p.mul_(8.0)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    content = torch.zeros(n_items + 1, 60, dtype=torch.int64)
    half = n_items // 2
    toks = torch.randint(5, 500, (half, 28), generator=g)
    for k in range(half):
        a, b = 2 * k + 1, 2 * k + 2
        content[a, 1:29] = toks[k]
        content[b, 1:29] = toks[k]
        j = torch.randint(0, 28, (3,), generator=g)
        content[b, 1 + j] = torch.randint(5, 500, (3,), generator=g)
    content[1:, 0], content[1:, 29], content[1:, 30:] = 101, 102, 1
    rng = np.random.default_rng(seed)
    eval_seq, hist = {}, {}
    for u in range(n_users):
        n = int(rng.integers(3, 22))
        seq = [int(x) for x in rng.choice(np.arange(1, n_items + 1), size=n, replace=False)]
        last = seq[-2]
        twin = last + 1 if last % 2 == 1 else last - 1
        if rng.random() < 0.7 and twin not in seq[:-1]:
            seq[-1] = twin
        eval_seq[u], hist[u] = seq, torch.LongTensor(seq[:-1])
    return model, args, content, eval_seq, hist


def test_eval_hr_ndcg_fp32_and_bf16_vs_oracle_2000_items():
    import test_engine_gpu as TG
    from adapter4rec_amd.data_utils import eval_model, get_item_embeddings
    from adapter4rec_amd.data_utils.metrics import eval_ranks
    from oracle import ref_cpu as R
    model, args, content, eval_seq, hist = build_eval_case()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = dict(R.DEFAULT_CFG, bert_heads=2)
    emb_ref = R.item_embeddings(sd, content.numpy(), cfg)
    users, ranks_ref = R.eval_ranks(sd, emb_ref, eval_seq, hist, cfg)
    hr_ref, nd_ref = R.hit_ndcg(ranks_ref)
    print(f'oracle: HR@10 {hr_ref:.4f} nDCG@10 {nd_ref:.4f}')
    assert hr_ref > 0.25                                          # the check below is not vacuous
    log = logging.getLogger('parity-eval')
    res = {}
    for dtype in ('fp32', 'bf16', 'bf16+fp32sweep'):
        model.compute_dtype = dtype[:4]
        args.eval_compute_dtype = 'fp32' if dtype.endswith('sweep') else None      # run.py's default: item sweep in fp32 on the bf16-trained weights
        model.invalidate_native()
        model.to(DEV)
        emb = get_item_embeddings(model, content.numpy(), 256, args, True, 0)
        hr = eval_model(model, hist, eval_seq, emb, 128, args, content.shape[0] - 1, log, 'test', 0)
        ranks = eval_ranks(model, hist, eval_seq, emb, 128, args, list(range(len(eval_seq)))).cpu().numpy()
        h2, nd = R.hit_ndcg(ranks)
        assert abs(h2 - hr) < 1e-6
        d = np.abs(ranks - ranks_ref)
        res[dtype] = dict(hr=hr, ndcg=nd, emb_err=float((emb.cpu() - emb_ref).abs().max()), same_rank=float((d == 0).mean()),
                          within_1=float((d <= 1).mean()), max_rank_diff=int(d.max()),
                          top10_flips=int(((ranks <= 10) != (ranks_ref <= 10)).sum()))
        print(dtype, res[dtype])
        model.cpu()
    f, b, m = res['fp32'], res['bf16'], res['bf16+fp32sweep']
    # fp32 instantiation, and the default eval of a bf16 training run (--eval_compute_dtype fp32: the item sweep on a forward-only
    # fp32 snapshot of the same weights): the north_star bar (HR@10 / nDCG@10 within 1e-3), ranks equal up to fp32 near-ties
    for r in (f, m):
        assert abs(r['hr'] - hr_ref) < 1e-3 and abs(r['ndcg'] - nd_ref) < 1e-3, r
        assert r['same_rank'] > 0.99 and r['max_rank_diff'] <= 2, r
    # bf16 item sweep (--eval_compute_dtype bf16): scores carry ~2^-8 relative error, so a few users whose target sits at the
    # rank-10 boundary flip.  Measured on MI355X: HR@10 0.4150 vs 0.4133, nDCG@10 0.2915 vs 0.2904, 3 of 600 users flipped.
    assert abs(b['hr'] - hr_ref) < 5e-3 and abs(b['ndcg'] - nd_ref) < 5e-3, b
    assert b['top10_flips'] <= 6 and b['within_1'] > 0.6, b
