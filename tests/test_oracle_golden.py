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
