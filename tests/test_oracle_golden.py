"""CPU: the oracle (oracle/ref_cpu.py) against golden vectors produced by the imported reference."""
import numpy as np
import pytest
import torch

from golden_util import GOLDEN, LRS, VARIANT_CFG, load_variant, strip
from oracle import ref_cpu as R

TOL = 1e-4   # scores / loss within 1e-4 abs on the fp32 path


@pytest.mark.parametrize('name', list(VARIANT_CFG))
def test_forward_and_grads(name):
    sd, cfg, fx, trainable, (items, mask), _ = load_variant(name)
    with torch.no_grad():
        out = R.model_forward(sd, items, mask, cfg)
    np.testing.assert_allclose(out['input_embs_all'].numpy(), fx['input_embs_all'], atol=TOL, rtol=0)
    np.testing.assert_allclose(out['prec_vec'].numpy(), fx['prec_vec'], atol=TOL, rtol=0)
    assert abs(float(out['loss']) - float(fx['loss'])) < TOL
    gkeys = [k[5:] for k in fx.files if k.startswith('grad/')]
    assert gkeys
    _, grads = R.loss_and_grads(sd, [strip(k) for k in gkeys], items, mask, cfg)
    for k in gkeys:
        ref = fx['grad/' + k]
        got = grads[strip(k)].numpy()
        np.testing.assert_allclose(got, ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('tower', ['text', 'image'])
def test_lora_pinned_through_merged_weights(tower):
    """a7: loralib==0.1.1 is absent, so LoRA is pinned ALGEBRAICALLY through the reference's own numbers (tools/gen_golden_r4.py): with
    W = W_base - B A / r (B != 0) the oracle's lora_linear must reproduce the imported reference's forward on the base weights, and its
    dA / dB must equal the chain rule through the reference's own dL/dW of the replaced query / value / w_Q / w_V Linears."""
    from golden_util import lora_pin_case
    sd, cfg, (items, mask), pin, expect = lora_pin_case(tower)
    assert sum('lora_A' in k for k in expect) >= 6
    with torch.no_grad():
        out = R.model_forward(sd, items, mask, cfg)
    np.testing.assert_allclose(out['input_embs_all'].numpy(), pin['input_embs_all'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out['prec_vec'].numpy(), pin['prec_vec'], atol=1e-5, rtol=0)
    assert abs(float(out['loss']) - float(pin['loss'])) < 1e-5
    names = list(expect)
    _, grads = R.loss_and_grads(sd, names, items, mask, cfg)
    for n in names:
        ref = expect[n].numpy()
        np.testing.assert_allclose(grads[n].numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=n)


def test_hidden_states_per_layer():
    sd, cfg, fx, _, (items, _), _ = load_variant('houlsby')
    with torch.no_grad():
        _, hs = R.text_encoder(sd, items[:8], cfg, return_all=True)
    np.testing.assert_allclose(torch.stack(hs).numpy(), fx['hidden_states'], atol=TOL, rtol=0)


@pytest.mark.parametrize('name', ['houlsby', 'pfeiffer', 'compacter', 'roberta_cpc_pfeiffer'])
def test_adam_three_steps(name):
    sd, cfg, fx, trainable, batch, _ = load_variant(name)
    losses1, p1 = R.train_steps(sd, trainable, [batch], cfg, LRS, 1)
    losses3, p3 = R.train_steps(sd, trainable, [batch], cfg, LRS, 3)
    np.testing.assert_allclose(losses3, fx['adam_losses'], atol=TOL, rtol=0)
    # which lr group a tensor lands in depends on the *module* name incl. the CompacterModel 'model.' prefix
    for k in fx['trainable']:
        k = str(k)
        for tag, got in (('adam1/', p1), ('adam3/', p3)):
            ref = fx[tag + k]
            np.testing.assert_allclose(got[strip(k)].numpy(), ref, rtol=1e-4, atol=1e-7, err_msg=tag + k)


def test_dataset_fixture():
    import random
    fx = np.load(GOLDEN + '/dataset.npz')
    lens = fx['seq_len']
    flat = fx['seq_flat']
    seqs, o = [], 0
    for n in lens:
        seqs.append([int(x) for x in flat[o:o + n]])
        o += n
    random.seed(int(fx['seed']))
    content = fx['item_content']
    for u, seq in enumerate(seqs):
        ids, mask = R.build_train_sample(seq, 200, 20, random)
        np.testing.assert_array_equal(content[ids], fx['sample_items'][u])
        np.testing.assert_array_equal(mask, fx['log_mask'][u])


def test_eval_fixture():
    sd, cfg, _, _, _, base = load_variant('houlsby')
    fx = np.load(GOLDEN + '/eval.npz')
    emb = R.item_embeddings(sd, base['item_content'], cfg, bs=64)
    np.testing.assert_allclose(emb.numpy(), fx['item_embeddings'], atol=TOL, rtol=0)
    seqs, o = {}, 0
    for u, n in enumerate(fx['full_seq_len']):
        seqs[u] = [int(x) for x in fx['full_seq_flat'][o:o + n]]
        o += n
    for tag in ('valid', 'test'):
        ev, hist = {}, {}
        for u, s in seqs.items():
            tr, va, te, hv, ht = R.split_sequences(s, 20)
            ev[u], hist[u] = (va, hv) if tag == 'valid' else (te, ht)
        users, ranks = R.eval_ranks(sd, emb, ev, hist, cfg)
        per_user = fx[tag + '_hit_ndcg_per_user'][:len(ranks)]   # SequentialDistributedSampler pads the tail (dataset.py:95-104)
        hit = (ranks <= 10).astype(np.float32)
        ndcg = np.where(ranks <= 10, 1.0 / np.log2(ranks + 1.0), 0.0)
        np.testing.assert_array_equal(hit, per_user[:, 0])
        np.testing.assert_allclose(ndcg, per_user[:, 1], atol=1e-6)
        hr, nd = R.hit_ndcg(ranks)
        assert abs(hr - float(fx[tag + '_means'][0])) < 1e-3      # HR@10 / nDCG@10 within 1e-3
        assert abs(nd - float(fx[tag + '_means'][1])) < 1e-3


# ------------------------------------------------------------------ image path (ViT / ViT-MAE tower)
from golden_util import CV_LRS, CV_VARIANT_CFG, load_cv_variant  # noqa: E402


@pytest.mark.parametrize('name', list(CV_VARIANT_CFG))
def test_cv_forward_and_grads(name):
    sd, cfg, fx, trainable, (images, mask), _ = load_cv_variant(name)
    with torch.no_grad():
        out = R.model_forward(sd, images, mask, cfg)
    np.testing.assert_allclose(out['input_embs_all'].numpy(), fx['input_embs_all'], atol=TOL, rtol=0)
    np.testing.assert_allclose(out['prec_vec'].numpy(), fx['prec_vec'], atol=TOL, rtol=0)
    assert abs(float(out['loss']) - float(fx['loss'])) < TOL * max(1.0, float(fx['loss']))
    gkeys = [k[5:] for k in fx.files if k.startswith('grad/')]
    assert gkeys or name == 'cv_vit_frozen'
    if gkeys:
        _, grads = R.loss_and_grads(sd, [strip(k) for k in gkeys], images, mask, cfg)
        for k in gkeys:
            ref = fx['grad/' + k]
            np.testing.assert_allclose(grads[strip(k)].numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('name', ['cv_vit_houlsby', 'cv_vit_compacter', 'cv_mae_houlsby', 'cv_vit_kadapter'])
def test_cv_adam_three_steps(name):
    sd, cfg, fx, trainable, batch, _ = load_cv_variant(name)
    # lr groups are decided on the module names incl. the CompacterModel 'model.' prefix (none of the tests here depends on it)
    losses3, p3 = R.train_steps(sd, trainable, [batch], cfg, CV_LRS, 3, group=R.lr_group_cv)
    np.testing.assert_allclose(losses3, fx['adam_losses'], atol=TOL * 20, rtol=0)
    for k in fx['trainable']:
        k = str(k)
        # atol: Adam's update is lr * m / (sqrt(v) + 1e-8); where |g| is at fp32-rounding level the ratio is rounding noise, so an element
        # may move by a small fraction of lr (5e-4 here) differently; 1e-6 = 0.2 % of one step
        np.testing.assert_allclose(p3[strip(k)].numpy(), fx['adam3/' + k], rtol=2e-4, atol=1e-6, err_msg=k)


def test_cv_backbone_matches_installed_hf():
    """The generator first checked its 4.20.1-shaped backbone against the installed HuggingFace ViT / ViT-MAE."""
    d = np.load(GOLDEN + '/cv_base.npz')['hf_check']
    assert d.max() < 2e-5


# ------------------------------------------------------------------ round 3: the oracle pinned at the BENCHMARKED geometry
BASE_GEOM = {
    'bert_houlsby_gelu': (dict(encoder='bert', act='GELU'), dict(adapter_activation='GELU')),
    'roberta_pfeiffer_cpc': (dict(encoder='roberta', act='relu', adapter_type='pfeiffer', arch='cpc'),
                             dict(adapter_activation='relu', adapter_type='pfeiffer', arch='cpc', encoder='roberta', bert_ln_eps=1e-5, pad_token_id=1)),
}


@pytest.mark.parametrize('name', list(BASE_GEOM))
def test_oracle_at_base_geometry_vs_imported_reference(name):
    """BERT-base (vocab 30 522) and RoBERTa-base (vocab 50 265, position ids offset by the pad id) geometry, 12 layers, 84 items: the
    restatement vs what the IMPORTED reference computed on the same seeded weights (tools/gen_golden_r3.py base): loss, scores,
    embeddings, prec_vec and the gradients the fixture keeps."""
    import os
    from base_cases import build_text_case, checksum
    kw, ocfg = BASE_GEOM[name]
    model, items, mask = build_text_case(**kw)
    fx = np.load(os.path.join(GOLDEN, f'base_geom_{name}.npz'))
    want = float(fx['weights_checksum'])
    assert abs(checksum(model) - want) <= 1e-6 * max(1.0, abs(want))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    kept = [k[5:] for k in fx.files if k.startswith('grad/')]
    out, grads = R.loss_and_grads(sd, kept, items, mask, dict(R.DEFAULT_CFG, **ocfg))
    assert abs(float(out['loss'].detach()) - float(fx['loss'])) < TOL
    np.testing.assert_allclose(out['input_embs_all'].detach().numpy(), fx['input_embs_all'], atol=TOL, rtol=0)
    np.testing.assert_allclose(out['prec_vec'].detach().numpy(), fx['prec_vec'], atol=TOL, rtol=0)
    pos, neg = out['pos_score'].detach().numpy(), out['neg_score'].detach().numpy()
    if ocfg.get('arch') == 'cpc':
        np.testing.assert_allclose(pos, fx['pos_score'][:, -1], atol=TOL, rtol=0)
        np.testing.assert_allclose(neg, fx['neg_score'][:, -1], atol=TOL, rtol=0)
    else:
        v = mask.bool().numpy()
        np.testing.assert_allclose(pos[v], fx['pos_score'][v], atol=TOL, rtol=0)
        np.testing.assert_allclose(neg[v], fx['neg_score'][v], atol=TOL, rtol=0)
    for k in kept:
        ref = fx['grad/' + k]
        np.testing.assert_allclose(grads[k].numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=k)


@pytest.mark.parametrize('name', ['houlsby', 'houlsby_gelu', 'pfeiffer', 'roberta_cpc_pfeiffer'])
def test_autocast_fixture_is_a_reduced_precision_run_of_the_same_step(name):
    """<name>_autocast.npz = the imported reference under torch.autocast(bfloat16) on the weights / batch of <name>.npz: close to the
    fp32 fixture (a bf16-sized distance, not a different computation) and not identical to it."""
    import os
    fx, ac = np.load(os.path.join(GOLDEN, name + '.npz')), np.load(os.path.join(GOLDEN, name + '_autocast.npz'))
    d_emb = np.abs(ac['input_embs_all'] - fx['input_embs_all']).max()
    assert 1e-5 < d_emb < 5e-2 and abs(float(ac['loss']) - float(fx['loss'])) < 5e-2
    rel = ac['grad_rel_err_vs_fp32']
    assert 1e-4 < rel.max() < 0.2
    keys = [k for k in ac.files if k.startswith('grad/')]
    assert len(keys) == len(rel) == len([k for k in fx.files if k.startswith('grad/')])


def test_multi_attribute_news_vs_reference():
    """--news_attributes title,abstract (Downstream/Text/model/encoders.py:60-99): the oracle's multi-attribute branch against the imported reference's
    own Bert_Encoder / Model (tests/golden/multi_attr.npz, tools/gen_golden_r5.py): per-attribute vectors, their mean, prec_vec, loss, every gradient."""
    from golden_util import load_multi_attr
    from oracle import ref_cpu as R
    sd, cfg, fx, trainable, (items, mask) = load_multi_attr()
    nt = int(fx['num_words'][0])
    with torch.no_grad():
        tv = R.text_encoder(sd, items[:, :2 * nt], dict(cfg, news_attributes=['title']))
        av = R.text_encoder(sd, items[:, 2 * nt:], dict(cfg, news_attributes=['abstract']))
    np.testing.assert_allclose(tv.numpy(), fx['title_vec'], atol=1e-4, rtol=0)
    np.testing.assert_allclose(av.numpy(), fx['abstract_vec'], atol=1e-4, rtol=0)
    out, grads = R.loss_and_grads(sd, trainable, items, mask, cfg)
    assert abs(float(out['loss'].detach()) - float(fx['loss'])) < 1e-4
    np.testing.assert_allclose(out['input_embs_all'].detach().numpy(), fx['input_embs_all'], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out['prec_vec'].detach().numpy(), fx['prec_vec'], atol=1e-4, rtol=0)
    for k in fx['trainable']:
        ref = fx['grad/' + str(k)]
        np.testing.assert_allclose(grads[strip(str(k))].numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0, err_msg=str(k))

