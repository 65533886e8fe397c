"""GPU: the whole native training step (Model.forward / backward / FusedAdam through the C ABI) against
(a) golden vectors produced by the imported reference and (b) the CPU oracle, on identical weights and batches.

fp32 compute mode is held to the north_star tolerance (scores / loss 1e-4 abs, gradients 1e-4 rel);
bf16 mode to a bf16-rounding bound that is written next to each assertion.
"""
import argparse

import numpy as np
import pytest
import torch

from golden_util import LRS, VARIANT_CFG, load_variant, strip

pytestmark = pytest.mark.gpu

GEOM = dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=40, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
            attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert')
ARGS = dict(houlsby=dict(), houlsby_gelu=dict(adapter_activation='GELU'), houlsby_parallel=dict(is_serial='None'),
            pfeiffer=dict(adapter_type='pfeiffer', adapter_activation='relu'), pfeiffer_ver2=dict(adapter_type='pfeiffer_ver2'),
            compacter=dict(adapter_type='compacter'), houlsby_cpc=dict(arch='cpc'), prompt=dict(adapter_type='prompt', n_tokens=8),
            kadapter=dict(adapter_type='kadapter', k_adapter_bert_list='0,1', k_adapter_bert_hidden_dim=64, num_adapter_heads_bert=4,
                          num_adapter_heads_sasrec=2),
            roberta_cpc_pfeiffer=dict(adapter_type='pfeiffer', adapter_activation='relu', arch='cpc', bert_model_load='roberta_tiny'),
            roberta_prompt=dict(adapter_type='prompt', n_tokens=8, arch='cpc', bert_model_load='roberta_tiny'))


def make_args(**kw):
    a = argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=128,
        bert_model_load='bert_tiny', bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation='RELU', hypercomplex_division=4, phm_init_range=1e-4, adapter_type='houslby', is_serial='True',
        adding_adapter_to='all', arch='sasrec', compute_dtype='fp32', **LRS)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def build(name, dtype='fp32'):
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model, ModelCPC
    sd, cfg, fx, trainable, (items, mask), base = load_variant(name)
    args = make_args(compute_dtype=dtype, **ARGS[name])
    geom = dict(GEOM)
    if name.startswith('roberta'):
        geom.update(max_position_embeddings=42, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, model_type='roberta')
    torch.manual_seed(0)
    bert = BertBackbone(geom)
    model = (ModelCPC if args.arch == 'cpc' else Model)(args, 200, True, bert)
    freeze_all(model)
    root = inject_adapters(model, args)
    full = {str(k): sd[strip(str(k))] for k in fx['all_keys']}
    root.load_state_dict(full, strict=True)           # the reference's own key names load unchanged
    names = {n for n, p in root.named_parameters() if p.requires_grad}
    assert names == {str(k) for k in fx['trainable']}, 'same trainable set as the reference'
    root.to('cuda:0')
    root.eval()                                       # dropout off: parity is defined without it
    return root, args, sd, cfg, fx, items.to('cuda:0'), mask.to('cuda:0')


@pytest.mark.parametrize('name', ['houlsby', 'houlsby_cpc', 'pfeiffer'])
def test_step_fp32_unused_item_slots_not_encoded(name, monkeypatch):
    """A4R_SKIP_UNUSED_ITEMS=2 forces the compact item batch (engine.py: _kept_rows -- the last negative of every user under SASRec, all negatives
    but one under CPC are never read by Model.forward / ModelCPC.forward and are not encoded): loss, scores and every trainable gradient still
    equal the reference's numbers; with =0 (every slot encoded) likewise -- the two paths agree."""
    res = {}
    for mode in ('2', '0'):
        monkeypatch.setenv('A4R_SKIP_UNUSED_ITEMS', mode)
        root, args, sd, cfg, fx, items, mask = build(name, 'fp32')
        inner = getattr(root, 'model', root)
        eng = inner._engine()
        assert (eng._kept_rows(mask.shape[0]) is not None) == (mode == '2')
        loss = root(items, mask, 0)
        loss.backward()
        assert abs(loss.item() - float(fx['loss'])) < 1e-4, (mode, loss.item(), float(fx['loss']))
        params = dict(root.named_parameters())
        for k in fx['trainable']:
            k = str(k)
            ref = fx['grad/' + k]
            np.testing.assert_allclose(params[k].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=f'{mode} {k}')
        res[mode] = (loss.item(), {str(k): params[str(k)].grad.clone() for k in fx['trainable']})
    assert abs(res['2'][0] - res['0'][0]) < 1e-5
    for k, g in res['2'][1].items():
        assert float((g - res['0'][1][k]).abs().max()) <= 1e-6 + 2e-5 * float(g.abs().max()), k


@pytest.mark.parametrize('name', ['houlsby', 'houlsby_cpc', 'compacter', 'kadapter', 'roberta_cpc_pfeiffer'])
def test_step_fp32_host_log_mask_pad_slots_not_encoded(name):
    """log_mask handed over on the HOST (what run.py does with the DataLoader's tensor): the engine reads the batch's pad structure from it without a
    device synchronisation and does not encode the item slots of short histories that the loss never reads (a4r_rows_idx_copy gathers the rest).  The
    fixtures' batches hold two short users of four: loss and every trainable gradient still equal the REFERENCE's numbers (which encoded every slot)."""
    root, args, sd, cfg, fx, items, mask = build(name, 'fp32')
    inner = getattr(root, 'model', root)
    eng = inner._engine()
    loss = root(items, mask.cpu(), 0)
    kidx = eng._ctx['kidx']
    assert kidx is not None and kidx[1] < items.shape[0] and eng._ctx['n_items'] == kidx[1]
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4, (loss.item(), float(fx['loss']))
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        k = str(k)
        ref = fx['grad/' + k]
        np.testing.assert_allclose(params[k].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer'])
def test_step_fp32_short_titles_run_on_fewer_tokens(name):
    """A batch handed over on the host whose titles all end by token 16: the training step runs on 16 tokens per item instead of 30 (pad tokens never
    reach the CLS output) -- loss and every gradient equal the oracle's on the same batch at the full title length; also together with the host
    log_mask (pad slots of the fixture's two short users left out)."""
    import test_engine_host_logic as TH
    root, items, mask, names, out, grads = TH.short_title_case(name, device='cuda:0')
    inner = getattr(root, 'model', root)
    loss = root(items, mask, 0)                      # both still CPU tensors: Model.forward uploads them
    c = inner._engine()._ctx
    assert c['S'] == 16 and c['kidx'] is not None
    loss.backward()
    assert abs(loss.item() - float(out['loss'].detach())) < 1e-4
    params = dict(root.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    emb = inner.bert_encoder(items.to('cuda:0'))
    assert inner._engine().S == items.shape[1] // 2 and emb.shape[0] == items.shape[0]


@pytest.mark.parametrize('name', list(ARGS))
def test_step_fp32_vs_reference_golden(name):
    root, args, sd, cfg, fx, items, mask = build(name, 'fp32')
    inner = getattr(root, 'model', root)
    loss = root(items, mask, 0)
    loss.backward()
    eng = inner._engine()
    # scores / loss within 1e-4 abs, checked against the reference's numbers and the oracle's
    assert abs(loss.item() - float(fx['loss'])) < 1e-4, (loss.item(), float(fx['loss']))
    from oracle import ref_cpu as R
    with torch.no_grad():
        out = R.model_forward(sd, items.cpu(), mask.cpu(), cfg)
    embs = inner.bert_encoder(items)
    np.testing.assert_allclose(embs.cpu().numpy(), fx['input_embs_all'], atol=1e-4, rtol=0)
    if cfg['arch'] != 'cpc':
        pos, neg = eng_scores(root, items, mask)
        np.testing.assert_allclose(pos.cpu().numpy(), out['pos_score'].numpy(), atol=1e-4, rtol=0)
        np.testing.assert_allclose(neg.cpu().numpy(), out['neg_score'].numpy(), atol=1e-4, rtol=0)
    # adapter gradients within 1e-4 relative (of the tensor's max) vs the reference's autograd
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        k = str(k)
        ref = fx['grad/' + k]
        got = params[k].grad.cpu().numpy()
        tol = 1e-6 + 1e-4 * np.abs(ref).max()
        np.testing.assert_allclose(got, ref, atol=tol, rtol=0, err_msg=k)


def eng_scores(root, items, mask):
    inner = getattr(root, 'model', root)
    eng = inner._engine()
    with torch.no_grad():
        eng.train_forward(items, mask)
        s = eng.scores()
        eng._ctx = None
    return s


@pytest.mark.parametrize('name', ['houlsby', 'pfeiffer', 'compacter'])
def test_fused_adam_three_steps_vs_reference(name):
    from adapter4rec_amd.inject import optimizer_groups
    from adapter4rec_amd.optim import FusedAdam
    root, args, sd, cfg, fx, items, mask = build(name, 'fp32')
    opt = FusedAdam(optimizer_groups(root, args))
    losses = []
    params = dict(root.named_parameters())
    for s in range(3):
        opt.zero_grad()
        loss = root(items, mask, 0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if s in (0, 2):
            for k in fx['trainable']:
                k = str(k)
                np.testing.assert_allclose(params[k].detach().cpu().numpy(), fx[f'adam{s + 1}/' + k], rtol=2e-4, atol=2e-7, err_msg=k)
    np.testing.assert_allclose(losses, fx['adam_losses'], atol=1e-4, rtol=0)


def condition(sd):
    """The golden weights are random-init + jitter and give |score| ~ 8 (loss ~ 8), where sigmoid saturates and any
    rounding is amplified exponentially.  Shrinking the item head brings scores to O(1) (loss ~ 1.4, the regime of a
    trained model), which is where a bf16-vs-fp32 bound is meaningful."""
    sd = dict(sd)
    for k in ('bert_encoder.text_encoders.title.fc.weight', 'bert_encoder.text_encoders.title.fc.bias'):
        sd[k] = sd[k] * 0.25
    return sd


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer'])
def test_step_bf16_bound(name):
    """bf16 storage / fp32 accumulate vs the fp32 oracle on well-conditioned weights.
    Bound: ~10 bf16 roundings (2^-9 relative each) per layer on 2 layers => item embeddings within 2e-2 abs,
    scores (|s| <~ 2) within 5e-2 abs, loss within 2e-2, adapter gradients within 15 % of each tensor's max
    (the worst tensors are bias gradients of the first layer, sums of a few hundred signed terms)."""
    from oracle import ref_cpu as R
    root, args, sd, cfg, fx, items, mask = build(name, 'bf16')
    sd = condition(sd)
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    trainable = [strip(str(k)) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, trainable, items.cpu(), mask.cpu(), cfg)
    loss = root(items, mask, 0)
    loss.backward()
    inner = getattr(root, 'model', root)
    embs = inner.bert_encoder(items).cpu()
    emb_err = (embs - out['input_embs_all'].detach()).abs().max().item()
    loss_err = abs(loss.item() - float(out['loss'].detach()))
    params = dict(root.named_parameters())
    worst, worst_k = 0.0, None
    for k in fx['trainable']:
        k = str(k)
        refg = grads[strip(k)].numpy()
        rel = np.abs(params[k].grad.cpu().numpy() - refg).max() / (np.abs(refg).max() + 1e-12)
        if rel > worst:
            worst, worst_k = rel, k
    print(f'bf16 {name}: loss {loss.item():.5f} vs {float(out["loss"].detach()):.5f}, emb err {emb_err:.2e}, worst grad rel {worst:.3f} ({worst_k})')
    assert emb_err < 2e-2, emb_err
    assert loss_err < 2e-2, loss_err
    assert worst < 0.15, (worst, worst_k)          # (0.11 - 0.14 on the user tower's 16-wide adapters, by which roundings the item tower's forward has)


def test_step_bf16_residual_fp32():
    """--residual_dtype fp32 (the item encoder's residual stream between sub-layers in fp32, as under the reference's autocast) on the tiny
    Houlsby fixture: the step stays inside the bf16 bounds of test_step_bf16_bound, the fp32 twins are really used (embeddings differ from the
    bf16-residual run) and the embeddings are at least as close to the fp32 oracle as with the bf16 stream."""
    from oracle import ref_cpu as R
    root, args, sd, cfg, fx, items, mask = build('houlsby', 'bf16')
    sd = condition(sd)
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    trainable = [strip(str(k)) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, trainable, items.cpu(), mask.cpu(), cfg)
    inner = getattr(root, 'model', root)
    res = {}
    for rd in ('bf16', 'fp32', 'bf24', 'bf20'):
        inner.args.residual_dtype = rd
        inner.invalidate_native()
        for p in root.parameters():
            p.grad = None
        loss = root(items, mask, 0)
        loss.backward()
        assert inner._engine().res32 == (rd == 'fp32') and inner._engine().res24 == (rd in ('bf24', 'bf20')) and inner._engine().lo_div == (2 if rd == 'bf20' else 1)
        emb = inner.bert_encoder(items).cpu()
        params = dict(root.named_parameters())
        worst = max(float(np.abs(params[str(k)].grad.cpu().numpy() - grads[strip(str(k))].numpy()).max() / (np.abs(grads[strip(str(k))].numpy()).max() + 1e-12))
                    for k in fx['trainable'])
        res[rd] = dict(loss=abs(loss.item() - float(out['loss'].detach())), emb=emb, emb_err=float((emb - out['input_embs_all'].detach()).abs().max()),
                       emb_rms=float((emb - out['input_embs_all'].detach()).double().pow(2).mean().sqrt()), grad=worst)
    print({k: {a: (round(b, 5) if isinstance(b, float) else None) for a, b in v.items()} for k, v in res.items()})
    r32, r16 = res['fp32'], res['bf16']
    assert r32['loss'] < 2e-2 and r32['emb_err'] < 2e-2 and r32['grad'] < 0.15, r32
    assert not torch.equal(r32['emb'], r16['emb'])
    assert r32['emb_rms'] <= 1.05 * r16['emb_rms'], (r32['emb_rms'], r16['emb_rms'])
    r24 = res['bf24']                                  # the 24-bit stream (one byte per element beside the bf16 tensor): what the fp32 twins buy
    assert r24['loss'] < 2e-2 and r24['emb_err'] < 2e-2 and r24['grad'] < 0.15, r24
    assert not torch.equal(r24['emb'], r16['emb']) and r24['emb_rms'] <= 1.02 * r32['emb_rms'] + 1e-6, (r24['emb_rms'], r32['emb_rms'])
    r20 = res['bf20']                                  # the 20-bit stream (a nibble per element): the same to within the 2^-13 it rounds at
    assert r20['loss'] < 2e-2 and r20['emb_err'] < 2e-2 and r20['grad'] < 0.15, r20
    assert not torch.equal(r20['emb'], r16['emb']) and r20['emb_rms'] <= 1.05 * r32['emb_rms'] + 1e-6, (r20['emb_rms'], r32['emb_rms'])


@pytest.mark.parametrize('name', ['houlsby', 'roberta_cpc_pfeiffer'])
def test_step_bf16_matches_bf16_restatement(name):
    """The same step through tests/sim_lib.py (a torch restatement of the kernels' semantics that rounds to bf16 at the
    same storage points) must agree closely: what is left is accumulation order inside the MFMA tiles."""
    import sim_lib
    import adapter4rec_amd.engine as E
    root, args, sd, cfg, fx, items, mask = build(name, 'bf16')
    sd = condition(sd)
    full = {str(k): sd[strip(str(k))] for k in fx['all_keys']}
    root.load_state_dict(full, strict=True)
    loss = root(items, mask, 0)
    loss.backward()
    g_gpu = {n: p.grad.cpu().clone() for n, p in root.named_parameters() if p.requires_grad}
    l_gpu = loss.item()
    real_L, real_req = E.L, E.TransRecEngine._require_device
    try:
        E.L = sim_lib
        E.TransRecEngine._require_device = lambda self, p0: None
        root.cpu()
        root.load_state_dict(full, strict=True)
        for p in root.parameters():
            p.grad = None
        loss_c = root(items.cpu(), mask.cpu(), 'cpu')
        loss_c.backward()
    finally:
        E.L, E.TransRecEngine._require_device = real_L, real_req
    assert abs(l_gpu - loss_c.item()) < 5e-3, (l_gpu, loss_c.item())
    worst, where, gated = 0.0, '', 0.0
    for n, p in root.named_parameters():
        if p.requires_grad:
            ref = p.grad.numpy()
            e = np.abs(g_gpu[n].numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
            # the user tower's ReLU-gated down-projections rest on a few dozen rows: one relu' that flips between the two implementations' roundings moves
            # them by 0.02 (bf16 stream) .. 0.07 (bf20) of their max on the same weights (tools/_probe/restate_check.py) -- bounded apart
            if 'user_encoder' in n and 'fc_down' in n:
                gated = max(gated, e)
            elif e > worst:
                worst, where = e, n
    print(f'bf16 HIP vs bf16 restatement ({name}): loss {l_gpu:.5f} vs {loss_c.item():.5f}, worst gradient {worst:.4f} of its tensor max ({where}); user-tower fc_down {gated:.4f}')
    assert worst < 0.04, (worst, where)      # (two bf16 implementations: 0.029 - 0.030 measured; not parity evidence, a plumbing check)
    assert gated < 0.15, gated


@pytest.mark.parametrize('name,dtype', [('houlsby', 'fp32'), ('roberta_cpc_pfeiffer', 'fp32'), ('houlsby_parallel', 'fp32'), ('houlsby', 'bf16')])
def test_step_titles_of_different_lengths_are_packed(name, dtype):
    """Titles of 3 .. 20 tokens handed over on the host: the item tower runs on the attended tokens only (packed rows, a4r_attn_t.offsets) -- about
    a third of the rectangular 30-token batch -- and loss and every gradient still equal the ORACLE on the rectangular batch (fp32: 1e-4; bf16: the
    bounds of test_step_bf16_bound on conditioned weights)."""
    from oracle import ref_cpu as R
    import test_engine_host_logic as HL
    root, args, sd, cfg, fx, items, mask = build(name, dtype)
    if dtype == 'bf16':
        sd = condition(sd)
        root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    items = items.cpu().clone()
    S = items.shape[1] // 2
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(3, 21, (items.shape[0],), generator=g)
    col = torch.arange(S)[None, :]
    items[:, :S] = torch.where(col < lens[:, None], items[:, :S], torch.full_like(items[:, :S], 1 if name.startswith('roberta') else 0))
    items[:, S:] = (col < lens[:, None]).long()
    names = [str(k) for k in fx['trainable']]
    out, grads = R.loss_and_grads(sd, [strip(k) for k in names], items, mask.cpu(), cfg)
    loss = root(items, mask, 0)                       # the id rows on the HOST (what run.py's DataLoader hands over)
    eng = getattr(root, 'model', root)._engine()
    pk = eng._ctx['pk']
    assert pk is not None and pk['Mtok'] < items.shape[0] * 21
    loss.backward()
    tol_l, tol_g = (1e-4, 1e-4) if dtype == 'fp32' else (2e-2, 0.15)
    assert abs(loss.item() - float(out['loss'].detach())) < tol_l, (loss.item(), float(out['loss'].detach()))
    params = dict(root.named_parameters())
    for k in names:
        ref = grads[strip(k)].numpy()
        err = np.abs(params[k].grad.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < tol_g, (k, err)


def test_dropout_training_mode_runs_and_is_seeded():
    root, args, sd, cfg, fx, items, mask = build('houlsby', 'bf16')
    root.train()
    l1 = root(items, mask, 0)
    l1.backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in root.parameters() if p.requires_grad]).clone()
    assert torch.isfinite(l1) and torch.isfinite(g1).all()
    assert abs(l1.item() - float(fx['loss'])) > 1e-3          # dropout changes the loss
    inner = getattr(root, 'model', root)
    inner._engine().step_count = 0                            # same seed => same masks => same loss
    for p in root.parameters():
        p.grad = None
    l2 = root(items, mask, 0)
    assert abs(l2.item() - l1.item()) < 1e-5                  # (the loss sum uses float atomics: last-bit order effects)


def test_no_cpu_path():
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model
    args = make_args()
    model = Model(args, 200, True, BertBackbone(GEOM))
    freeze_all(model)
    model = inject_adapters(model, args)
    with pytest.raises(RuntimeError):
        model(torch.zeros(42, 60, dtype=torch.long), torch.ones(1, 20), 'cpu')


def test_eval_pipeline_vs_reference_fixture():
    """a13 on the GPU: get_item_embeddings + eval_model (a4r_eval_rank) vs the reference's HR@10 / nDCG@10 (abs 1e-3)."""
    import logging
    import os
    from golden_util import GOLDEN
    from oracle import ref_cpu as R
    from adapter4rec_amd.data_utils import eval_model, get_item_embeddings
    from adapter4rec_amd.data_utils.metrics import eval_ranks
    root, args, sd, cfg, fx0, items, mask = build('houlsby', 'fp32')
    base = np.load(os.path.join(GOLDEN, 'base.npz'))
    fx = np.load(os.path.join(GOLDEN, 'eval.npz'))
    emb = get_item_embeddings(root, base['item_content'], 64, args, True, 0)
    np.testing.assert_allclose(emb.cpu().numpy(), fx['item_embeddings'], atol=1e-4, rtol=0)
    seqs, o = {}, 0
    for u, n in enumerate(fx['full_seq_len']):
        seqs[u] = [int(x) for x in fx['full_seq_flat'][o:o + n]]
        o += n
    log = logging.getLogger('t')
    for tag in ('valid', 'test'):
        ev, hist = {}, {}
        for u, s in seqs.items():
            tr, va, te, hv, ht = R.split_sequences(s, 20)
            ev[u], hist[u] = (va, torch.tensor(hv)) if tag == 'valid' else (te, torch.tensor(ht))
        hr = eval_model(root, hist, ev, emb, 16, args, 200, log, tag, 0)
        assert abs(hr - float(fx[tag + '_means'][0])) < 1e-3
        ranks = eval_ranks(root, hist, ev, emb, 16, args, list(range(len(seqs)))).cpu().numpy()
        users, oracle_ranks = R.eval_ranks(sd, torch.from_numpy(fx['item_embeddings']), ev, hist, cfg)
        assert int(np.abs(ranks - oracle_ranks).sum()) <= 1          # bit-exact up to one fp32 near-tie
        nd = np.where(ranks <= 10, 1.0 / np.log2(ranks + 1.0), 0.0)
        assert abs(nd.mean() - float(fx[tag + '_means'][1])) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('tower', ['text', 'image'])
def test_lora_step_vs_reference_merged(tower):
    """a7 through the C ABI, pinned by the reference's own numbers: LoRA layers whose merged weight is the base weight reproduce the imported
    reference's loss (1e-4) and their dA / dB follow the reference's dL/dW (tests/golden/lora_pin_*.npz, tools/gen_golden_r4.py)."""
    from test_engine_host_logic import lora_pin_step
    lora_pin_step('cuda:0', tower)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_lora_vs_oracle(dtype):
    """a7: LoRA on q/v (BERT) and w_Q/w_V (SASRec).  loralib is third party and absent => parity unpinned by the reference;
    the native path is held to the oracle's restatement (fp32: 1e-4; bf16: the bound of test_step_bf16_bound)."""
    from oracle import ref_cpu as R
    from test_engine_host_logic import build_lora_cpu
    model, args, osd, ocfg, items, mask = build_lora_cpu(dtype)
    if dtype == 'bf16':
        osd = condition(osd)
        model.load_state_dict(osd, strict=True)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out, grads = R.loss_and_grads(osd, names, items, mask, ocfg)
    model.to('cuda:0').eval()
    loss = model(items.to('cuda:0'), mask.to('cuda:0'), 0)
    loss.backward()
    tol_l, tol_g = (1e-4, 1e-4) if dtype == 'fp32' else (2e-2, 0.12)
    assert abs(loss.item() - float(out['loss'].detach())) < tol_l
    params = dict(model.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, atol=1e-6 + tol_g * np.abs(ref).max(), rtol=0, err_msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize('ln_only', [False, True])
def test_finetune_all_fp32_vs_oracle_and_fixture(ln_only):
    """--fine_tune_to all (every backbone weight, the embedding tables, the item head and the SASRec weights trainable) and
    --finetune_layernorm without adapters, through the C ABI: every gradient vs the oracle's autograd and the reference fixture."""
    import test_engine_host_logic as TH
    from oracle import ref_cpu as R
    model, args, sd, cfg, fx, items, mask = TH.build_finetune_all(device='cuda:0', ln_only=ln_only)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    out, grads = R.loss_and_grads(sd, names, items.cpu(), mask.cpu(), cfg)
    loss = model(items, mask, 'cuda:0')
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    params = dict(model.named_parameters())
    for n in names:
        ref = grads[n].numpy()
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)
    if not ln_only:
        from adapter4rec_amd.inject import optimizer_groups
        from adapter4rec_amd.optim import FusedAdam
        opt = FusedAdam(optimizer_groups(model, args))
        l0 = loss.item()
        for _ in range(3):
            opt.zero_grad()
            l_ = model(items, mask, 'cuda:0')
            l_.backward()
            opt.step()
        assert model(items, mask, 'cuda:0').item() < l0          # three Adam steps on one batch must lower its loss


@pytest.mark.gpu
def test_finetune_all_bf16_large_weight_gradient_kernel():
    """--fine_tune_to all at a geometry where the backbone's weight gradients take the 256 x 256-tile kernel (H = 768, F = 3072, 5 376 token rows:
    a4r_gemm_tn256.hip through a4r_gemm_tn_bias / a4r_gemm_tn_multi -- q, k, v and the attention output in one launch, bias sums in the launch):
    every gradient of a 2-layer bf16 step against the same step on the 64-tile kernels + a4r_colsum (a4r_gemm_variant(6)).  Both sum the same
    bf16 products in fp32; only the order differs."""
    import bench
    from adapter4rec_amd import _lib as L
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    args = bench.make_args(4, 'bf16')
    args.adapter_type, args.adding_adapter_to, args.drop_rate = 'none', 'None', 0.0
    torch.manual_seed(5)
    model = Model(args, 2000, True, BertBackbone(dict(BERT_BASE, num_hidden_layers=2, vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)))
    for n, p in model.named_parameters():
        p.requires_grad = 'pooler' not in n
    model.to('cuda:0').eval()
    g = torch.Generator().manual_seed(6)
    content = bench.synth_content(2000, g)
    content[1:, 1:29] = torch.randint(1000, 1999, (2000, 28), generator=g)
    items, mask = [t.to('cuda:0') for t in bench.synth_batches(content, 2000, 4, 1, g)[0]]
    grads = []
    for variant in (7, 6):
        L.gemm_variant(variant)
        model.zero_grad()
        loss = model(items, mask, 'cuda:0')
        loss.backward()
        L.gemm_variant(7)
        grads.append({n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 30 and grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert float(b.abs().max()) > 0 or 'position_embeddings' in n or 'token_type' in n, n
        assert float((a - b).abs().max()) <= 1e-6 + 2e-4 * float(b.abs().max()), (n, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize('name', ['houlsby', 'houlsby_gelu', 'pfeiffer', 'roberta_cpc_pfeiffer'])
def test_step_bf16_vs_reference_autocast(name):
    """the bf16 path is bounded by the reference's OWN reduced-precision path.  <name>_autocast.npz
    (tools/gen_golden_r3.py) holds the imported reference's step under torch.autocast(bfloat16) (`with autocast(): bz_loss = model(...)`,
    Pretraining/Text/run.py:319-324) on the weights and batch of <name>.npz.  Both distances are measured from the SAME fp32 reference
    numbers: the HIP bf16 step may be at most 2x as far from them as the reference's autocast step is (plus a small floor for
    quantities the autocast run happens to hit exactly)."""
    import os
    from golden_util import GOLDEN
    root, args, sd, cfg, fx, items, mask = build(name, 'bf16')
    ac = np.load(os.path.join(GOLDEN, name + '_autocast.npz'))
    loss = root(items, mask, 0)
    loss.backward()
    inner = getattr(root, 'model', root)
    embs = inner.bert_encoder(items).cpu().numpy()
    params = dict(root.named_parameters())
    d_hip = dict(loss=abs(loss.item() - float(fx['loss'])), emb=np.abs(embs - fx['input_embs_all']).max())
    d_ac = dict(loss=abs(float(ac['loss']) - float(fx['loss'])), emb=np.abs(ac['input_embs_all'] - fx['input_embs_all']).max())
    g_hip, g_ac = [], []
    for k in fx['trainable']:
        k = str(k)
        ref = fx['grad/' + k]
        s = np.abs(ref).max() + 1e-30
        g_hip.append(np.abs(params[k].grad.cpu().numpy() - ref).max() / s)
        g_ac.append(np.abs(ac['grad/' + k] - ref).max() / s)
    d_hip['grad'], d_ac['grad'] = max(g_hip), max(g_ac)
    # per tensor the ratio is noisy (a bias gradient is a signed sum of a few hundred terms, and the worst tensor of one run is not the
    # worst tensor of the other): the median and the 90th percentile over tensors are the stable figures, the worst tensor gets 3x
    ratio = np.array(g_hip) / (np.array(g_ac) + 1e-12)
    med, p90 = float(np.median(ratio)), float(np.percentile(ratio, 90))
    worst_k = str(fx['trainable'][int(np.argmax(g_hip))])
    print(f'{name}: HIP bf16 vs fp32 reference {d_hip}; reference autocast(bf16) vs fp32 reference {d_ac}; per-tensor gradient error ratio: '
          f'median {med:.2f}, 90th percentile {p90:.2f}; worst HIP tensor {worst_k}')
    assert d_hip['emb'] <= 2.0 * d_ac['emb'] + 1e-3, (d_hip, d_ac)
    assert d_hip['loss'] <= 2.0 * d_ac['loss'] + 2e-2, (d_hip, d_ac)          # (a scalar: signed errors cancel in either run -- the autocast run's 0.003 on the CPC case is such a cancellation)
    assert med <= 2.0 and p90 <= 3.0, (med, p90)
    assert d_hip['grad'] <= 3.0 * d_ac['grad'] + 5e-3 and d_hip['grad'] < 0.12, (d_hip, d_ac)


def _text_fp8_case(dev, adapter_type='houslby', act='GELU'):
    """fp8 encoder on the post-LN text tower at a geometry where every e4m3 GEMM engages (H = 256: qkv [768, 256], attention-output
    [256, 256], FFN [512, 256] / [256, 512]; 2 layers, 4 heads of 64): fp8 vs bf16 vs the fp32 oracle on the same conditioned weights.
    Reference op: HF BertLayer inside Downstream/Text/model/encoders.py:39-56 (the reference has no reduced-precision text path)."""
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model
    from oracle import ref_cpu as R
    geom = dict(GEOM, hidden_size=256, num_attention_heads=4, intermediate_size=512)
    args = make_args(compute_dtype='fp32', word_embedding_dim=256, bert_model_load='bert_mini', adapter_type=adapter_type, adapter_activation=act)      # ('mini': width 256, model.py:23)
    torch.manual_seed(7)
    model = Model(args, 60, True, BertBackbone(geom))
    freeze_all(model)
    model = inject_adapters(model, args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn_like(p))
            if n.endswith('title.fc.weight') or n.endswith('title.fc.bias'):
                p.mul_(0.25)                                  # O(1) scores (see condition())
    model.eval()
    g = torch.Generator().manual_seed(2)
    B, Lh, S = 3, 21, 30
    ids = torch.zeros(B, Lh, 2, 2 * S, dtype=torch.int64)
    mask = torch.zeros(B, Lh - 1)
    for u in range(B):
        n = Lh if u == 0 else 6 + 5 * u
        for slot in range(Lh - n, Lh):
            for side in range(2):
                if side == 1 and slot == Lh - 1:
                    continue
                ln = int(torch.randint(5, S + 1, (1,), generator=g))
                ids[u, slot, side, :ln] = torch.randint(1, 120, (ln,), generator=g)
                ids[u, slot, side, S:S + ln] = 1
        mask[u, Lh - n:] = 1
    items = ids.view(-1, 2 * S)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    cfg = dict(R.DEFAULT_CFG, adapter_type=adapter_type, adapter_activation=act, bert_heads=4)
    out, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    ref_loss = float(out['loss'].detach())
    res = {}
    for dtype in ('bf16', 'fp8'):
        model.compute_dtype = dtype
        model.invalidate_native()
        for p in model.parameters():
            p.grad = None
        model.to(dev)
        loss = model(items.to(dev), mask.to(dev), dev)
        loss.backward()
        if dtype == 'fp8':
            eng = model._engine()
            assert eng.fp8 and all(b.wqkv8 is not None and b.wo8 is not None and b.wi8 is not None and b.wo28 is not None and b.wo2T8 is not None
                                   for b in eng.bert_blocks)
        with torch.no_grad():
            emb = model.bert_encoder(items.to(dev)).cpu()
        worst = 0.0
        for n, p in model.named_parameters():
            if p.requires_grad:
                r = grads[n]
                worst = max(worst, float((p.grad.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)))
        res[dtype] = (abs(float(loss.detach()) - ref_loss), float((emb - out['input_embs_all'].detach()).abs().max()), worst)
        model.cpu()
    print(f'text fp8 encoder vs fp32 oracle ({adapter_type}): loss / emb / worst-gradient error bf16 {res["bf16"]}, fp8 {res["fp8"]} (loss {ref_loss:.4f})')
    assert res['bf16'][0] < 3e-2 and res['bf16'][1] < 3e-2 and res['bf16'][2] < 0.15, res
    assert res['fp8'][0] < 8e-2 and res['fp8'][1] < 0.1 and res['fp8'][2] < 0.4, res
    return res


@pytest.mark.parametrize('adapter_type', ['houslby', 'pfeiffer'])
def test_text_fp8_encoder_gpu(adapter_type):
    _text_fp8_case('cuda:0', adapter_type)


def build_multi_attr(device, dtype='fp32'):
    """the reference's --news_attributes title,abstract fixture (tests/golden/multi_attr.npz) loaded into this package's classes"""
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BertBackbone, Model
    from golden_util import load_multi_attr
    sd, cfg, fx, trainable, (items, mask) = load_multi_attr()
    args = make_args(compute_dtype=dtype, news_attributes=['title', 'abstract'], num_words_title=int(fx['num_words'][0]), num_words_abstract=int(fx['num_words'][1]))
    torch.manual_seed(0)
    model = Model(args, 200, True, BertBackbone(dict(GEOM)))
    freeze_all(model)
    root = inject_adapters(model, args)
    root.load_state_dict({str(k): sd[strip(str(k))] for k in fx['all_keys']}, strict=True)
    assert {n for n, p in root.named_parameters() if p.requires_grad} == {str(k) for k in fx['trainable']}
    root.to(device)
    root.eval()
    return root, args, fx, items.to(device), mask.to(device)


@pytest.mark.gpu
def test_step_fp32_news_attributes_vs_reference_golden():
    """--news_attributes title,abstract (encoders.py:60-99; round 5): the engine's attribute stacking against the imported reference's own numbers --
    item embeddings (training batch and inference entry point), prec_vec, loss 1e-4; every adapter gradient 1e-4 x its max."""
    root, args, fx, items, mask = build_multi_attr('cuda:0')
    loss = root(items, mask, 0)
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) < 1e-4
    real = (items != 0).any(1).cpu().numpy()           # (the PAD item's never-read vector: DESIGN section 7)
    emb = root.bert_encoder(items).cpu().numpy()
    np.testing.assert_allclose(emb[real], fx['input_embs_all'][real], atol=1e-4, rtol=0)
    params = dict(root.named_parameters())
    for k in fx['trainable']:
        ref = fx['grad/' + str(k)]
        np.testing.assert_allclose(params[str(k)].grad.cpu().numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=str(k))

