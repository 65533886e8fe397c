"""The reference's own shipped data (VERDICT r5 missing #2 / weak #7; fixtures by tools/gen_golden_r6.py from the IMPORTED reference):

  CPU  * the repo's readers (text read_news_bert / get_doc_input_bert / read_behaviors, image read_images / read_behaviors) on the committed
         heads of Dataset/Adressa/Adressa_news_base.tsv and Dataset/Amazon/amazon_2w_{items,users}.tsv -- and on the FULL files wherever the
         reference tree is present -- against the SHA-256 of what the reference's readers returned (bit-equal: index work);
       * BuildTrainDataset on the 32 real histories of the batch fixture under the fixture's python-random seed: the same item ids;
       * the CPU oracle on the real batch (real Adressa titles, real Amazon history lengths, BERT-mini geometry, real 30 522 vocabulary)
         against the reference's loss / scores / embeddings / gradients: 1e-4.
  GPU  * the HIP fp32 instantiation through the C ABI on the same batch, handed over as the DataLoader would (host tensors: titles packed,
         pad slots found in the host mask and not encoded): 1e-4 against the REFERENCE's numbers; the bf16 instantiation inside its bounds
         and against the reference's own autocast(bfloat16) distance.
"""
import json
import logging
import os
import random

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, sha_array, sha_mapping
from real_data_util import head_paths

REF = '/root/reference'
FULL = dict(news=os.path.join(REF, 'Dataset/Adressa/Adressa_news_base.tsv'), items=os.path.join(REF, 'Dataset/Amazon/amazon_2w_items.tsv'),
            users=os.path.join(REF, 'Dataset/Amazon/amazon_2w_users.tsv'),
            vocab_dir=os.path.join(REF, 'Downstream/Text/pretrained_models/bert/bert_base_uncased'))
CLEAR = 64


def _run_repo_readers(paths):
    import argparse
    from transformers import BertTokenizer
    from adapter4rec_amd.cv import data_utils as CVD
    from adapter4rec_amd.data_utils import get_doc_input_bert, read_behaviors, read_news_bert
    log = logging.getLogger('real')
    args = argparse.Namespace(news_attributes=['title'], num_words_title=30, num_words_abstract=50, num_words_body=50)
    tok = BertTokenizer.from_pretrained(paths['vocab_dir'])
    id2dic, name2id = read_news_bert(paths['news'], args, tok)
    title, mask, a1, a2, b1, b2 = get_doc_input_bert(id2dic, args)
    assert a1 is None and a2 is None and b1 is None and b2 is None
    keys, img_name2id = CVD.read_images(paths['items'])
    cvr = CVD.read_behaviors(paths['users'], keys, img_name2id, 20, 5, log)
    txr = read_behaviors(paths['users'], {i: n for n, i in img_name2id.items()}, img_name2id, 20, 5, log)
    h = dict(news_name_to_id=sha_mapping(name2id), news_title=sha_array(title), news_title_attmask=sha_array(mask),
             images_name_to_id=sha_mapping(img_name2id), images_id_to_keys=sha_mapping(keys))
    for tag, r in (('cv', cvr), ('text', txr)):
        h[f'{tag}_item_num'] = int(r[0])
        for nm, d in zip(('item_id_to', 'users_train', 'users_valid', 'users_test', 'history_valid', 'history_test'), r[1:]):
            h[f'{tag}_{nm}'] = sha_mapping(d)
    return h, title, mask, cvr


def _check_readers(tag, paths):
    fx = np.load(os.path.join(GOLDEN, 'real_readers.npz'))
    want = json.loads(str(fx[f'{tag}/hashes']))
    got, title, mask, cvr = _run_repo_readers(paths)
    # the clear heads first (a readable failure), then every hash
    assert title.dtype == np.int32 and mask.dtype == np.int32                      # preprocess.py:114-115
    np.testing.assert_array_equal(title[:CLEAR + 1], fx[f'{tag}/title_head'])
    np.testing.assert_array_equal(mask[:CLEAR + 1], fx[f'{tag}/mask_head'])
    assert len(cvr[2]) == int(fx[f'{tag}/n_users'])
    for u in range(CLEAR):
        assert ' '.join(map(str, cvr[2][u])) == str(fx[f'{tag}/train_head'][u]), u
        assert ' '.join(map(str, cvr[4][u])) == str(fx[f'{tag}/test_head'][u]), u
    assert isinstance(cvr[5][0], torch.Tensor) and cvr[5][0].dtype == torch.int64   # preprocess.py:58-59: LongTensor histories
    bad = {k: (got[k], want[k]) for k in want if got.get(k) != want[k]}
    assert not bad, bad
    return got


def test_readers_on_committed_heads_equal_reference_hashes():
    """First 1 024 Adressa news + first 1 024 Amazon users (min_seq_len 5 drops some; items renumbered by occurrence): bit-equal."""
    with head_paths() as hp:
        got = _check_readers('head', hp)
    assert 0 < got['cv_item_num'] < 14720


@pytest.mark.skipif(not os.path.exists(FULL['users']), reason='the reference tree (its Dataset/ files) is not on this machine')
def test_readers_on_full_shipped_files_equal_reference_hashes():
    """All 20 373 Adressa titles (30 522-entry vocabulary) and all 21 153 Amazon users: every returned array / dict hashes as the reference's."""
    got = _check_readers('full', FULL)
    assert got['cv_item_num'] == 14430 and got['text_item_num'] == 14430


def test_real_shapes_file_is_the_readers_histogram():
    """tests/golden/real_shapes.json (what `bench.py --real-shaped` draws from) against the committed heads: the histogram format and that a
    head of the data has the same character (short titles, very short histories)."""
    with open(os.path.join(GOLDEN, 'real_shapes.json')) as f:
        sh = json.load(f)
    t, h = sh['title_tokens'], sh['history_items']
    assert len(t['values']) == len(t['counts']) == 31 and len(h['values']) == len(h['counts']) == 22
    assert sum(t['counts']) == 20373 and sum(h['counts']) == 21153
    assert abs(np.dot(t['values'], t['counts']) / sum(t['counts']) - t['mean']) < 1e-9 and 11 < t['mean'] < 12
    assert abs(np.dot(h['values'], h['counts']) / sum(h['counts']) - h['mean']) < 1e-9 and 4 < h['mean'] < 4.3
    assert h['counts'][0] == h['counts'][1] == h['counts'][2] == 0                  # min_seq_len 5 -> train histories of >= 3 items
    with head_paths() as hp:
        _, _, mask, cvr = _run_repo_readers(hp)
    assert abs(mask[1:].sum(1).mean() - t['mean']) < 1.0
    assert abs(np.mean([len(s) for s in cvr[2].values()]) - h['mean']) < 0.5


def _fixture():
    return np.load(os.path.join(GOLDEN, 'real_batch.npz'))


def _batch(fx):
    items = torch.from_numpy(fx['sample_items'].astype(np.int64))
    return items, torch.from_numpy(fx['log_mask'])


def test_build_train_dataset_on_real_histories_draws_the_reference_batch():
    """The repo's BuildTrainDataset on the fixture's 32 real Amazon histories under its python-random seed: the ids the reference's class drew
    (dataset.py:24-49), and the token rows are those ids' rows of the content table."""
    from adapter4rec_amd.data_utils import BuildTrainDataset
    fx = _fixture()
    seqs, o = {}, 0
    for u, n in enumerate(fx['seq_len']):
        seqs[u] = [int(x) for x in fx['seq_flat'][o:o + n]]
        o += n
    ds = BuildTrainDataset(u2seq=seqs, item_content=None, item_num=int(fx['item_num']), max_seq_len=20, use_modal=False)
    random.seed(int(fx['seed']))
    ids = torch.stack([ds[u][0] for u in range(len(seqs))])
    np.testing.assert_array_equal(ids.numpy(), fx['sample_ids'])
    masks = torch.stack([ds[u][1] for u in range(len(seqs))])
    np.testing.assert_array_equal(masks.numpy(), fx['log_mask'])
    # pad slots carry the PAD item = all-zero token rows; every other slot a title with [CLS] first
    rows = fx['sample_items'].reshape(32, 21, 2, 60)
    pad = fx['sample_ids'] == 0
    assert (rows[pad] == 0).all() and (rows[~pad][:, 0] == 101).all() and (rows[~pad][:, 30] == 1).all()


def _reference_numbers(fx, mask):
    valid = mask.bool()
    return dict(loss=float(fx['loss']), pos=torch.from_numpy(fx['pos_score'])[valid], neg=torch.from_numpy(fx['neg_score'])[valid],
                emb=torch.from_numpy(fx['input_embs_all']), grads={k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('grad/')})


def _build(fx):
    from base_cases import build_real_case, checksum
    model = build_real_case(int(fx['item_num']))
    got, want = checksum(model), float(fx['weights_checksum'])
    assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), f'seeded weights differ from the fixture generator ({got} vs {want})'
    return model


def _slots_read(fx):
    """Item slots Model.forward reads (model.py:48-70): a positive that is an input or a target, a negative under a set mask."""
    m = fx['log_mask'] > 0
    pos = np.zeros((m.shape[0], 21), bool)
    pos[:, :-1] |= m
    pos[:, 1:] |= m
    neg = np.zeros((m.shape[0], 21), bool)
    neg[:, :-1] = m
    return np.stack([pos, neg], axis=2).reshape(-1)


def test_oracle_on_real_batch_vs_reference():
    """oracle/ref_cpu.py on real titles / real history lengths vs the imported reference: loss, scores, embeddings 1e-4, every gradient 1e-4 of
    its tensor's max (GELU adapters)."""
    from oracle import ref_cpu as R
    torch.set_num_threads(8)
    fx = _fixture()
    model = _build(fx)
    items, mask = _batch(fx)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == [str(n) for n in fx['names']]
    out, grads = R.loss_and_grads(sd, names, items, mask, dict(R.DEFAULT_CFG, adapter_activation='GELU', bert_heads=4))
    ref = _reference_numbers(fx, mask)
    valid = mask.bool()
    assert abs(float(out['loss'].detach()) - ref['loss']) < 1e-4
    assert float((out['pos_score'].detach()[valid] - ref['pos']).abs().max()) < 1e-4 and float((out['neg_score'].detach()[valid] - ref['neg']).abs().max()) < 1e-4
    read = torch.from_numpy(_slots_read(fx))
    assert float((out['input_embs_all'].detach() - ref['emb'])[read].abs().max()) < 1e-4
    assert float((out['input_embs_all'].detach() - ref['emb']).abs().max()) < 1e-4           # the oracle encodes the pad item as the reference does
    for n in names:
        e = float((grads[n] - ref['grads'][n]).abs().max() / ref['grads'][n].abs().max().clamp_min(1e-30))
        assert e < 1e-4, (n, e)


@pytest.mark.gpu
def test_real_batch_step_hip_vs_reference():
    """HIP through the C ABI on the real batch, host tensors in (the DataLoader's hand-over: titles PACKED to their own lengths, pad slots found
    in the host mask and not encoded).  fp32: 1e-4 against the reference's own numbers; bf16: the stated bounds and the reference-under-autocast
    comparison of the base-geometry test."""
    from test_parity_base_gpu import grad_err, hip_step
    fx = _fixture()
    model = _build(fx)
    items, mask = _batch(fx)
    ref = _reference_numbers(fx, mask)
    valid = mask.bool()
    read = torch.from_numpy(_slots_read(fx))
    o = hip_step(model, 'fp32', items, mask, host=True)
    n_tok = int(fx['sample_items'].reshape(-1, 60)[read.numpy()][:, 30:].sum())
    print(f'real batch: {int(read.sum())} of {read.numel()} item slots read, {n_tok} attended tokens of {read.numel() * 30} rectangular '
          f'({n_tok / (read.numel() * 30):.3f}); the item tower ran on {o["packed_tokens"]} token rows')
    assert o['packed_tokens'] is not None and o['packed_tokens'] <= n_tok + 64, 'the titles were not packed / pad slots were encoded'

    d = dict(loss=abs(o['loss'] - ref['loss']), pos=float((o['pos'][valid] - ref['pos']).abs().max()), neg=float((o['neg'][valid] - ref['neg']).abs().max()),
             emb=float((o['emb'] - ref['emb'])[read].abs().max()))
    g, where = grad_err(o['grads'], ref['grads'])
    print('real batch fp32 HIP vs the imported reference:', d, 'worst gradient', g, where)
    assert d['loss'] < 1e-4 and d['pos'] < 1e-4 and d['neg'] < 1e-4 and d['emb'] < 1e-4, d
    assert g < 1e-4, (g, where)
    # the same step with every slot encoded at the full title length (device tensors in): the two hand-overs agree
    o2 = hip_step(model, 'fp32', items, mask, host=False)
    assert abs(o2['loss'] - o['loss']) < 2e-5 and grad_err(o2['grads'], o['grads'])[0] < 1e-4
    b = hip_step(model, 'bf16', items, mask, host=True)
    db = dict(loss=abs(b['loss'] - ref['loss']), pos=float((b['pos'][valid] - ref['pos']).abs().max()), emb=float((b['emb'] - ref['emb'])[read].abs().max()))
    gb, wb = grad_err(b['grads'], ref['grads'])
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    ac = dict(pos=rms(torch.from_numpy(fx['ac_pos_score'])[valid] - ref['pos']), emb=rms((torch.from_numpy(fx['ac_input_embs_all']) - ref['emb'])[read]))
    hp = dict(pos=rms(b['pos'][valid] - ref['pos']), emb=rms((b['emb'] - ref['emb'])[read]))
    print('real batch bf16 HIP vs the imported reference:', db, 'worst gradient', gb, wb)
    print('   rms distance from the fp32 reference, HIP bf16 vs reference under autocast(bfloat16):', hp, ac, {k: round(hp[k] / ac[k], 3) for k in hp})
    assert db['loss'] < 3e-2 and db['emb'] < 4e-2 and db['pos'] < 0.15 and gb < 0.2, (db, gb, wb)
    assert hp['pos'] <= 1.6 * ac['pos'] + 1e-3 and hp['emb'] <= 1.6 * ac['emb'] + 1e-3
