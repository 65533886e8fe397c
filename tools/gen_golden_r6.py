#!/usr/bin/env python
"""Round-6 fixtures: the reference's own READERS on the data files it ships, and one training step of the imported reference on a batch of REAL
titles and REAL histories (CPU, build container only; nothing here runs on the GPU box).

  python tools/gen_golden_r6.py readers     # tests/golden/real_readers.npz + tests/golden/real_data/* + tests/golden/real_shapes.json
  python tools/gen_golden_r6.py batch       # tests/golden/real_batch.npz
  python tools/gen_golden_r6.py cv_autocast # tests/golden/cv_vit_houlsby_autocast.npz

readers -- IMPORTS Downstream/Text/data_utils/preprocess.py (read_news_bert, get_doc_input_bert, read_behaviors: :5-63, :80-151) and
  Downstream/CV/data_utils/preprocess.py (read_images, read_behaviors: :5-83) and runs them on
    Dataset/Adressa/Adressa_news_base.tsv   (20 373 real news titles) with the shipped bert_base_uncased vocabulary, --num_words_title 30
    Dataset/Amazon/amazon_2w_items.tsv + amazon_2w_users.tsv   (14 720 items, 21 153 real user histories), --max_seq_len 20 --min_seq_len 5
  and stores the SHA-256 of every returned array / dict (tests/golden_util.py: sha_array, sha_mapping) plus the first 64 rows / users in clear,
  once for the FULL files (checked wherever /root/reference exists) and once for the committed heads of the files (tests/golden/real_data/:
  the first 1 024 news lines, the first 1 024 users, the item list, the vocabulary -- data, not source; checked everywhere).
  real_shapes.json: the histograms of attended tokens per title and of training-history lengths (what `bench.py --real-shaped` draws from).
batch -- 32 real Amazon users (a seeded draw: the real length distribution) whose item k reads Adressa title k, sampled by the reference's
  BuildTrainDataset under a python-random seed, through the IMPORTED reference Model (HF BertModel at bert_mini geometry with the real 30 522
  vocabulary, Houlsby GELU adapters) on the seeded weights of tests/base_cases.build_real_case: loss, scores, embeddings, prec_vec, every
  trainable gradient.
cv_autocast -- the image tower's reference wrappers under torch.autocast(bfloat16) (VERDICT r5 housekeeping: the text tower has *_autocast.npz,
  the image tower had none): written by tools/gen_golden_cv.py's harness, see gen_cv_autocast below.
"""
import gzip
import importlib.util
import json
import logging
import os
import random
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

REFROOT = '/root/reference'
ADRESSA = os.path.join(REFROOT, 'Dataset/Adressa/Adressa_news_base.tsv')
AMAZON_ITEMS = os.path.join(REFROOT, 'Dataset/Amazon/amazon_2w_items.tsv')
AMAZON_USERS = os.path.join(REFROOT, 'Dataset/Amazon/amazon_2w_users.tsv')
VOCAB_DIR = os.path.join(REFROOT, 'Downstream/Text/pretrained_models/bert/bert_base_uncased')
HEAD = 1024
CLEAR = 64
MAX_SEQ_LEN, MIN_SEQ_LEN, NW = 20, 5, 30


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_readers():
    """The reference's two preprocess modules, imported by path (both are called `preprocess`; neither imports anything but numpy / torch)."""
    text = _load(os.path.join(REFROOT, 'Downstream/Text/data_utils/preprocess.py'), 'ref_text_preprocess')
    cv = _load(os.path.join(REFROOT, 'Downstream/CV/data_utils/preprocess.py'), 'ref_cv_preprocess')
    return text, cv


def reader_args():
    import argparse
    return argparse.Namespace(news_attributes=['title'], num_words_title=NW, num_words_abstract=50, num_words_body=50)


def run_readers(text, cv, news_path, items_path, users_path, tokenizer, log):
    """-> dict of hashes + clear heads, the arrays themselves (for the batch generator)."""
    from golden_util import sha_array, sha_mapping
    args = reader_args()
    id2dic, name2id = text.read_news_bert(news_path, args, tokenizer)
    title, mask, a1, a2, b1, b2 = text.get_doc_input_bert(id2dic, args)
    assert a1 is None and a2 is None and b1 is None and b2 is None
    keys, img_name2id = cv.read_images(items_path)
    cvr = cv.read_behaviors(users_path, keys, img_name2id, MAX_SEQ_LEN, MIN_SEQ_LEN, log)
    # the text reader on the same behaviours file (same item list, ids as the values of its id -> content dict)
    txr = text.read_behaviors(users_path, {i: n for n, i in img_name2id.items()}, img_name2id, MAX_SEQ_LEN, MIN_SEQ_LEN, log)
    out = {}
    out['news_name_to_id'] = sha_mapping(name2id)
    out['news_title'], out['news_title_attmask'] = sha_array(title), sha_array(mask)
    out['images_name_to_id'], out['images_id_to_keys'] = sha_mapping(img_name2id), sha_mapping(keys)
    for tag, r in (('cv', cvr), ('text', txr)):
        out[f'{tag}_item_num'] = int(r[0])
        for nm, d in zip(('item_id_to', 'users_train', 'users_valid', 'users_test', 'history_valid', 'history_test'), r[1:]):
            out[f'{tag}_{nm}'] = sha_mapping(d)
    clear = dict(title_head=title[:CLEAR + 1].copy(), mask_head=mask[:CLEAR + 1].copy(),
                 train_head=np.array([' '.join(map(str, cvr[2][u])) for u in range(CLEAR)]),
                 test_head=np.array([' '.join(map(str, cvr[4][u])) for u in range(CLEAR)]),
                 n_users=np.array(len(cvr[2])), n_news=np.array(len(name2id)))
    return out, clear, dict(title=title, mask=mask, cv=cvr)


def write_heads(dst):
    os.makedirs(dst, exist_ok=True)
    with open(ADRESSA) as f, open(os.path.join(dst, 'adressa_news_head.tsv'), 'w') as g:
        for i, line in enumerate(f):
            if i >= HEAD:
                break
            g.write(line)
    with open(AMAZON_USERS) as f, open(os.path.join(dst, 'amazon_users_head.tsv'), 'w') as g:
        for i, line in enumerate(f):
            if i >= HEAD:
                break
            g.write(line)
    with open(AMAZON_ITEMS, 'rb') as f, gzip.GzipFile(os.path.join(dst, 'amazon_items.tsv.gz'), 'wb', mtime=0) as g:
        shutil.copyfileobj(f, g)
    with open(os.path.join(VOCAB_DIR, 'vocab.txt'), 'rb') as f, gzip.GzipFile(os.path.join(dst, 'bert_base_uncased_vocab.txt.gz'), 'wb', mtime=0) as g:
        shutil.copyfileobj(f, g)


def gen_readers():
    from transformers import BertTokenizer
    from golden_util import GOLDEN
    from real_data_util import head_paths
    text, cv = ref_readers()
    log = logging.getLogger('gen_r6')
    tok = BertTokenizer.from_pretrained(VOCAB_DIR)                       # Downstream/Text/run.py:96,298
    write_heads(os.path.join(GOLDEN, 'real_data'))
    out = {}
    full_h, full_c, full = run_readers(text, cv, ADRESSA, AMAZON_ITEMS, AMAZON_USERS, tok, log)
    with head_paths() as hp:
        tok_h = BertTokenizer.from_pretrained(hp['vocab_dir'])
        head_h, head_c, _ = run_readers(text, cv, hp['news'], hp['items'], hp['users'], tok_h, log)
    for tag, h, c in (('full', full_h, full_c), ('head', head_h, head_c)):
        out[f'{tag}/hashes'] = np.array(json.dumps(h, sort_keys=True))
        for k, v in c.items():
            out[f'{tag}/{k}'] = v
    np.savez_compressed(os.path.join(GOLDEN, 'real_readers.npz'), **out)
    # the real shapes: attended tokens per title (items 1 ..), training-history lengths (len(users_train[u]) = inputs + the last target)
    tl = full['mask'][1:].sum(1)
    hl = np.array([len(s) for s in full['cv'][2].values()])
    shapes = dict(source='Dataset/Adressa/Adressa_news_base.tsv titles (bert_base_uncased, --num_words_title 30); '
                         'Dataset/Amazon/amazon_2w_users.tsv users_train lengths (--max_seq_len 20, --min_seq_len 5)',
                  title_tokens=dict(values=list(range(NW + 1)), counts=np.bincount(tl, minlength=NW + 1).tolist(), mean=float(tl.mean())),
                  history_items=dict(values=list(range(MAX_SEQ_LEN + 2)), counts=np.bincount(hl, minlength=MAX_SEQ_LEN + 2).tolist(), mean=float(hl.mean())))
    with open(os.path.join(GOLDEN, 'real_shapes.json'), 'w') as f:
        json.dump(shapes, f, indent=1)
    print('readers: full', {k: (v if isinstance(v, int) else v[:12]) for k, v in full_h.items()})
    print(f"real shapes: title tokens mean {shapes['title_tokens']['mean']:.2f} max {int(tl.max())}; train-history items mean {shapes['history_items']['mean']:.2f} "
          f"of {MAX_SEQ_LEN + 1} slots; users {len(hl)}, news {len(tl)}")
    return full


def gen_batch():
    import base_cases as BC
    import gen_golden as G                                               # imports the reference's model / data_utils (Downstream/Text)
    import gen_golden_r3 as G3
    from golden_util import GOLDEN
    from transformers import BertConfig, BertModel, BertTokenizer
    text, cv = ref_readers()
    log = logging.getLogger('gen_r6')
    tok = BertTokenizer.from_pretrained(VOCAB_DIR)
    _, _, full = run_readers(text, cv, ADRESSA, AMAZON_ITEMS, AMAZON_USERS, tok, log)
    item_num, users_train = int(full['cv'][0]), full['cv'][2]
    content = np.concatenate([full['title'], full['mask']], axis=1)[:item_num + 1].astype(np.int64)      # Amazon item k reads Adressa title k
    users = sorted(int(u) for u in np.random.default_rng(606).choice(len(users_train), size=32, replace=False))
    ds = G.BuildTrainDataset(u2seq=users_train, item_content=content, item_num=item_num, max_seq_len=MAX_SEQ_LEN, use_modal=True)
    seed = 20260
    random.seed(seed)
    items, masks = zip(*[ds[u] for u in users])
    items, masks = torch.stack(items), torch.stack(masks)
    sample = items.view(-1, 2 * NW)

    model = BC.build_real_case(item_num)
    a = model.args
    args = G.make_args(word_embedding_dim=256, bert_model_load='bert_mini_uncased', adapter_activation=a.adapter_activation, adapter_type=a.adapter_type, arch=a.arch)
    geo = BC.GEOMETRY['bert']
    cfg = BertConfig(vocab_size=geo['vocab_size'], max_position_embeddings=geo['max_position_embeddings'], type_vocab_size=geo['type_vocab_size'],
                     layer_norm_eps=geo['layer_norm_eps'], pad_token_id=geo['pad_token_id'], attn_implementation='eager', **BC.REAL_MINI)
    ref = G.Model(args, item_num, True, BertModel(cfg))
    for p in ref.parameters():
        p.requires_grad = False
    ref = G.inject(ref, args)
    missing, unexpected = ref.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()}, strict=False)
    assert not unexpected and all('position_ids' in k or 'token_type_ids' in k for k in missing), (missing, unexpected)
    ref.eval()
    f32 = G3.run_once(ref, sample, masks, False)
    ac = G3.run_once(ref, sample, masks, True)
    names = list(f32['grads'])
    out = dict(users=np.array(users), seed=np.array(seed), item_num=np.array(item_num), sample_items=sample.numpy().astype(np.int32), log_mask=masks.numpy(),
               seq_len=np.array([len(users_train[u]) for u in users]), seq_flat=np.concatenate([np.array(users_train[u]) for u in users]),
               weights_checksum=np.array(BC.checksum(model)), names=np.array(names), loss=np.array(f32['loss']),
               pos_score=f32['pos'].numpy(), neg_score=f32['neg'].numpy(), input_embs_all=f32['emb'].numpy(), prec_vec=f32['prec'].numpy(),
               ac_loss=np.array(ac['loss']), ac_pos_score=ac['pos'].numpy(), ac_neg_score=ac['neg'].numpy(), ac_input_embs_all=ac['emb'].numpy(),
               ac_grad_rel_err=np.array([G3.rel_err(ac['grads'][n], f32['grads'][n]) for n in names]))
    # the item ids behind the token rows (the reference's dataset with use_modal=False draws the same negatives under the same seed)
    ds_ids = G.BuildTrainDataset(u2seq=users_train, item_content=None, item_num=item_num, max_seq_len=MAX_SEQ_LEN, use_modal=False)
    random.seed(seed)
    ids = torch.stack([ds_ids[u][0] for u in users])
    assert torch.equal(torch.from_numpy(content)[ids], items)
    out['sample_ids'] = ids.numpy().astype(np.int32)
    for n in names:
        out['grad/' + n] = f32['grads'][n].numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'real_batch.npz'), **out)
    valid = masks.bool()
    tl = sample[:, NW:].sum(1)
    print(f"real batch: 32 users, {int(valid.sum())} scored positions of {valid.numel()}, history items {out['seq_len'].tolist()}; attended tokens per item slot mean "
          f"{float(tl.float().mean()):.2f} (pad slots count 0; non-pad mean {float(tl[tl > 0].float().mean()):.2f}, max {int(tl.max())}); loss {f32['loss']:.6f} (autocast {ac['loss']:.6f}); "
          f"{len(names)} gradients; max |score| {float(f32['pos'][valid].abs().max()):.2f}")


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('readers', 'all'):
        gen_readers()
    if what in ('batch', 'all'):
        gen_batch()
    if what in ('cv_autocast', 'all'):
        import gen_golden_cv_autocast
        gen_golden_cv_autocast.main()
