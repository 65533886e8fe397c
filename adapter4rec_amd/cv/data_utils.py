"""Host-side data utilities of the image path with the reference's names (Downstream/CV/data_utils/preprocess.py,
metrics.py): read_images, read_behaviors, get_itemLMDB_embeddings; eval_model is the text path's (same contract:
metrics.py:85-119 == Downstream/Text/data_utils/metrics.py:82-116) and runs on a4r_eval_rank."""
import os
import pickle

import numpy as np
import torch

from ..data_utils.metrics import _gather_shards, _inner, _my_shard, eval_model, eval_ranks       # noqa: F401
from .image_io import RecordStore, decode_record, resize_to_square


def read_images(images_path):                                          # preprocess.py:70-83
    item_id_to_keys, item_name_to_id, index = {}, {}, 1
    with open(images_path, 'r') as f:
        for line in f:
            name = line.strip('\n').split('\t')[0]
            item_name_to_id[name] = index
            item_id_to_keys[index] = name.encode('ascii')
            index += 1
    return item_id_to_keys, item_name_to_id


def read_behaviors(behaviors_path, before_item_id_to_keys, before_item_name_to_id, max_seq_len, min_seq_len, Log_file):
    """preprocess.py:5-67: keep the last max_seq_len + 3 items of users with >= min_seq_len, renumber the items that occur,
    train = seq[:-2], valid = seq[-(L+2):-1], test = seq[-(L+1):]."""
    before_item_num = len(before_item_name_to_id)
    counts = [0] * (before_item_num + 1)
    user_seq_dic = {}
    with open(behaviors_path, 'r') as f:
        for line in f:
            sp = line.strip('\n').split('\t')
            names = sp[1].split(' ')
            if len(names) < min_seq_len:
                continue
            ids = [before_item_name_to_id[i] for i in names[-(max_seq_len + 3):]]
            user_seq_dic[sp[0]] = ids
            for i in ids:
                counts[i] += 1
    item_id_to_keys, before_to_now, item_id = {}, {}, 1
    for b in range(1, before_item_num + 1):
        if counts[b] != 0:
            before_to_now[b] = item_id
            item_id_to_keys[item_id] = before_item_id_to_keys[b]
            item_id += 1
    item_num = len(before_to_now)
    users_train, users_valid, users_test, hist_valid, hist_test = {}, {}, {}, {}, {}
    for uid, (_, seqs) in enumerate(user_seq_dic.items()):
        seq = [before_to_now[i] for i in seqs]
        users_train[uid] = seq[:-2]
        users_valid[uid] = seq[-(max_seq_len + 2):-1]
        users_test[uid] = seq[-(max_seq_len + 1):]
        hist_valid[uid] = torch.LongTensor(np.array(seq[:-2]))
        hist_test[uid] = torch.LongTensor(np.array(seq[:-1]))
    Log_file.info('##### items after clearing {}, user seqs {} #####'.format(item_num, len(users_train)))
    return item_num, item_id_to_keys, users_train, users_valid, users_test, hist_valid, hist_test


class _LmdbTxn:
    def __init__(self, path):
        import lmdb                                                     # dataset.py:67-69 (not in this image; used where it exists)
        self.env = lmdb.open(path, subdir=os.path.isdir(path), readonly=True, lock=False, readahead=False, meminit=False)

    def get(self, key):
        with self.env.begin() as txn:
            return txn.get(key)


def open_image_db(path):
    """``--lmdb_data``: an LMDB directory / file (dataset.py:69-74: through the ``lmdb`` module where it is installed, else through
    ``cv/lmdb_reader.py``, a read-only reader of the same file format) or a pickled RecordStore (``*.pkl``: the same key -> pickled
    LMDB_Image records)."""
    if path.endswith('.pkl'):
        with open(path, 'rb') as f:
            return RecordStore(pickle.load(f))
    try:
        return _LmdbTxn(path)
    except ImportError:
        from .lmdb_reader import LmdbReader
        return LmdbReader(path)


def get_itemLMDB_embeddings(model, item_num, item_id_to_keys, test_batch_size, args, local_rank, db=None):
    """metrics.py:68-82: encode items 0..item_num (0 = the all-zero pad image, dataset.py:163-166) -> fp32 [N + 1, E] on the device."""
    model.eval()
    db = db if db is not None else open_image_db(os.path.join(args.root_data_dir, args.dataset, args.lmdb_data))
    enc = _inner(model, args).cv_encoder
    ed = getattr(args, 'eval_compute_dtype', None)
    if ed and ed != _inner(model, args).compute_dtype:
        enc = _inner(model, args).item_encoder_in(ed)
    dev = next(model.parameters()).device
    R = args.CV_resize
    out = []
    lo, hi, chunk, world = _my_shard(item_num + 1)         # the sweep is sharded over the data-parallel ranks and all-gathered
    with torch.no_grad():
        for s in range(lo, hi, test_batch_size):
            ids = list(range(s, min(s + test_batch_size, hi)))
            batch = torch.zeros(len(ids), R, R, 3, dtype=torch.uint8, device=dev)
            groups = {}
            for j, i in enumerate(ids):
                if i == 0:
                    continue
                a = decode_record(db.get(item_id_to_keys[i]))
                groups.setdefault(a.shape, []).append((j, a))
            for shape, lst in groups.items():
                raw = torch.from_numpy(np.stack([a for _, a in lst])).to(dev)
                batch[torch.tensor([j for j, _ in lst], device=dev)] = resize_to_square(raw, R)
            if ids[0] == 0:                    # the reference's pad item is an all-ZERO float image, not a black uint8 one
                f = ((batch.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2).contiguous()
                f[0] = 0
                out.append(enc(f))
            else:
                out.append(enc(batch))
    mine = torch.cat(out, 0) if out else torch.zeros(0, int(args.embedding_dim), device=dev)
    return _gather_shards(mine, item_num + 1, chunk, world)
