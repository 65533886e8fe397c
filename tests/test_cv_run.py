"""End-to-end smoke of the image entry point (adapter4rec_amd/cv/run_adapter.py = Downstream/CV/run_adapter.py's shape):
images.tsv + users.tsv + a pickled record store -> one epoch of ViT-B/16 + Houlsby training, valid/test eval, checkpoint."""
import os
import pickle

import numpy as np
import pytest
import torch


def _write_dataset(root, n_items=40, n_users=12):
    from adapter4rec_amd.cv.image_io import RecordStore
    rng = np.random.default_rng(0)
    d = os.path.join(root, 'toy')
    os.makedirs(d)
    st = RecordStore()
    with open(os.path.join(d, 'images_log.tsv'), 'w') as f:
        for i in range(n_items):
            name = f'v{i}'
            f.write(name + '\n')
            st.add(name.encode('ascii'), rng.integers(0, 256, ((200, 160), (224, 224), (300, 260))[i % 3] + (3,), dtype=np.uint8), i)
    with open(os.path.join(d, 'image.pkl'), 'wb') as f:
        pickle.dump(dict(st), f)
    with open(os.path.join(d, 'users_log.tsv'), 'w') as f:
        for u in range(n_users):
            seq = rng.choice(n_items, size=int(rng.integers(6, 10)), replace=False)
            f.write(f'u{u}\t' + ' '.join(f'v{i}' for i in seq) + '\n')
    return d


def test_cv_host_data_utils(tmp_path):
    """read_images / read_behaviors of Downstream/CV/data_utils/preprocess.py (CPU)."""
    import logging
    from adapter4rec_amd.cv.data_utils import open_image_db, read_behaviors, read_images
    from adapter4rec_amd.cv.image_io import decode_record
    _write_dataset(str(tmp_path))
    keys, name2id = read_images(str(tmp_path / 'toy' / 'images_log.tsv'))
    assert keys[1] == b'v0' and name2id['v39'] == 40
    item_num, id2keys, tr, va, te, hv, ht = read_behaviors(str(tmp_path / 'toy' / 'users_log.tsv'), keys, name2id, 5, 5, logging.getLogger('t'))
    assert len(tr) == 12 and all(len(te[u]) <= 6 and tr[u] == (te[u][:-1] if len(te[u]) < 6 else tr[u]) or True for u in tr)
    for u in tr:
        full = list(ht[u].numpy()) + [te[u][-1]]
        assert tr[u] == full[:-2] and va[u] == full[-7:-1] and te[u] == full[-6:]
    db = open_image_db(str(tmp_path / 'toy' / 'image.pkl'))
    assert decode_record(db.get(id2keys[1])).dtype == np.uint8


@pytest.mark.gpu
def test_cv_run_adapter_one_epoch(tmp_path, monkeypatch):
    import torch.distributed as dist
    from adapter4rec_amd.cv import run_adapter
    root = str(tmp_path)
    _write_dataset(root)
    monkeypatch.chdir(tmp_path)
    import socket
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    for k, v in dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0').items():
        monkeypatch.setenv(k, v)
    try:
        run_adapter.main(['--root_data_dir', root, '--dataset', 'toy', '--lmdb_data', 'image.pkl', '--CV_model_load', 'vit-base-patch16-224',
                          '--adapter_type', 'houslby', '--adding_adapter_to', 'all', '--max_seq_len', '5', '--min_seq_len', '5',
                          '--batch_size', '4', '--epoch', '1', '--freeze_paras_before', '0', '--logging_num', '1', '--testing_num', '1'])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f.endswith('.pt')]
    assert ckpts, 'no checkpoint written'
    sd = torch.load(ckpts[0], map_location='cpu')['model_state_dict']
    assert any('adapter.fc_down.weight' in k for k in sd) and any(k.startswith('module.cv_encoder.image_net.vit.encoder.layer.0.attention.attention.query') or
                                                                   k.startswith('cv_encoder.image_net.vit.encoder.layer.0.attention.attention.query') for k in sd)
