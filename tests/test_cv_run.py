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
                          '--batch_size', '4', '--epoch', '1', '--freeze_paras_before', '0', '--logging_num', '1', '--testing_num', '1', '--num_workers', '0'])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs if f.endswith('.pt')]
    assert ckpts, 'no checkpoint written'
    sd = torch.load(ckpts[0], map_location='cpu')['model_state_dict']
    assert any('adapter.fc_down.weight' in k for k in sd) and any(k.startswith('module.cv_encoder.image_net.vit.encoder.layer.0.attention.attention.query') or
                                                                   k.startswith('cv_encoder.image_net.vit.encoder.layer.0.attention.attention.query') for k in sd)


# ---------------------------------------------------------------------------------------------------------------------------------------
# The image entry point at the text one's standard (tests/test_text_run.py): tiny ViT / ViT-MAE geometries read from config.json, two epochs +
# resume = the uninterrupted run, the logged HR@10 = the CPU oracle's on the saved weights, ViT-MAE, and Pretraining/CV -> Downstream/CV.
# Reference: Downstream/CV/run_adapter.py:284-636, Pretraining/CV/run.py:93-285, Downstream/CV/data_utils/dataset.py:85-113.
TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, image_size=32, patch_size=8)
N_ITEMS_T, N_USERS_T = 60, 40


def _write_tiny(root):
    """<root>/pretrained_models/{vit-base-patch16-224, vit-mae-base}/config.json (tiny geometries); <root>/data/toy/{images_log.tsv, users_log.tsv,
    image.pkl} with 32 x 32 uint8 records (a few larger ones: resized on the GPU); <root>/work (cwd: the entry point reads ../pretrained_models)."""
    import json
    from adapter4rec_amd.cv.image_io import RecordStore
    rng = np.random.default_rng(1)
    for name, extra in (('vit-base-patch16-224', {}), ('vit-mae-base', dict(mask_ratio=0.75))):
        d = os.path.join(root, 'pretrained_models', name)
        os.makedirs(d)
        with open(os.path.join(d, 'config.json'), 'w') as f:
            json.dump(dict(TINY, **extra), f)
    d = os.path.join(root, 'data', 'toy')
    os.makedirs(d)
    st = RecordStore()
    with open(os.path.join(d, 'images_log.tsv'), 'w') as f:
        for i in range(N_ITEMS_T):
            name = f'v{i}'
            f.write(name + '\n')
            st.add(name.encode('ascii'), rng.integers(0, 256, (32, 32, 3), dtype=np.uint8), i)
    with open(os.path.join(d, 'image.pkl'), 'wb') as f:
        pickle.dump(dict(st), f)
    with open(os.path.join(d, 'users_log.tsv'), 'w') as f:
        for u in range(N_USERS_T):
            seq = rng.choice(N_ITEMS_T, size=int(rng.integers(5, 26)), replace=False)
            f.write(f'u{u}\t' + ' '.join(f'v{i}' for i in seq) + '\n')
    os.makedirs(os.path.join(root, 'work'))
    return os.path.join(root, 'data')


def _run_cv(argv, monkeypatch, record):
    import logging
    import socket
    import torch.distributed as dist
    from adapter4rec_amd.cv import run_adapter as RA
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    for k, v in dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0').items():
        monkeypatch.setenv(k, v)
    orig_fwd, orig_eval = RA.FlatDDP.forward, RA.run_eval_once

    def fwd(self, *a, **k):
        out = orig_fwd(self, *a, **k)
        record['loss'].append(float(out.detach()))
        record['batch'].append(int(a[1].shape[0]))
        return out

    def ev(model, db, item_id_to_keys, user_history, users_eval, batch_size, item_num, mode, local_rank, args, Log_file):
        hit = orig_eval(model, db, item_id_to_keys, user_history, users_eval, batch_size, item_num, mode, local_rank, args, Log_file)
        record['eval'].append((mode, float(hit)))
        return hit
    monkeypatch.setattr(RA.FlatDDP, 'forward', fwd)
    monkeypatch.setattr(RA, 'run_eval_once', ev)
    try:
        RA.main(argv)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        monkeypatch.setattr(RA.FlatDDP, 'forward', orig_fwd)
        monkeypatch.setattr(RA, 'run_eval_once', orig_eval)
        for name in ('Log_file', 'Log_screen'):
            lg = logging.getLogger(name)
            for h in list(lg.handlers):
                lg.removeHandler(h)
                h.close()


def _simulate_cv(monkeypatch):
    import torch.distributed as dist
    import sim_lib
    import adapter4rec_amd.cv.image_io as IO
    import adapter4rec_amd.data_utils.metrics as MT
    import adapter4rec_amd.engine as E
    import adapter4rec_amd.engine_vit as EV
    import adapter4rec_amd.optim as O
    for mod in (E, EV, O, MT, IO):
        monkeypatch.setattr(mod, 'L', sim_lib)
    monkeypatch.setattr(E.TransRecEngine, '_require_device', lambda self, p0: None)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'get_rng_state', lambda *a: torch.zeros(1, dtype=torch.uint8))
    monkeypatch.setattr(torch.cuda, 'manual_seed_all', lambda s: None)
    real_init = dist.init_process_group
    monkeypatch.setattr(dist, 'init_process_group', lambda backend=None, **k: real_init('gloo', **k))
    def dev_cpu(x):
        if isinstance(x, bool):
            return x
        if isinstance(x, int) or (isinstance(x, str) and x.startswith('cuda')) or (isinstance(x, torch.device) and x.type == 'cuda'):
            return 'cpu'
        return x
    on_cpu = lambda a: [dev_cpu(x) for x in a]
    kw_cpu = lambda k: {q: (dev_cpu(v) if q == 'device' else v) for q, v in k.items() if q != 'non_blocking'}
    real_mto, real_tto = torch.nn.Module.to, torch.Tensor.to
    monkeypatch.setattr(torch.nn.Module, 'to', lambda self, *a, **k: real_mto(self, *on_cpu(a), **k))
    monkeypatch.setattr(torch.Tensor, 'to', lambda self, *a, **k: real_tto(self, *on_cpu(a), **kw_cpu(k)))
    for fn in ('zeros', 'tensor', 'empty', 'ones', 'arange', 'full'):
        real = getattr(torch, fn)
        monkeypatch.setattr(torch, fn, (lambda real: lambda *a, **k: real(*a, **{q: (dev_cpu(v) if q == 'device' else v) for q, v in k.items()}))(real))


COMMON_CV = ['--num_workers', '0', '--dataset', 'toy', '--lmdb_data', 'image.pkl', '--CV_resize', '32', '--freeze_paras_before', '0', '--embedding_dim', '64',
             '--batch_size', '16', '--logging_num', '3', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5', '--drop_rate', '0',
             '--adapter_dropout_rate', '0', '--compute_dtype', 'fp32']


def _oracle_hr(sd, data, va, hv, item_id_to_keys, mae=False):
    from oracle import ref_cpu as R
    from adapter4rec_amd.cv.data_utils import open_image_db
    from adapter4rec_amd.cv.image_io import decode_record
    db = open_image_db(os.path.join(data, 'toy', 'image.pkl'))
    n = max(item_id_to_keys) + 1
    imgs = torch.zeros(n, 3, 32, 32)
    for i in range(1, n):
        imgs[i] = R.normalize_u8(torch.from_numpy(np.array(decode_record(db.get(item_id_to_keys[i]))))[None])[0]
    cfg = dict(R.DEFAULT_CFG, tower='image', vit_heads=2, mae=mae)
    osd = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        emb = torch.cat([R.image_encoder(osd, imgs[i:i + 32], cfg) for i in range(0, n, 32)], 0)
    _, ranks = R.eval_ranks(osd, emb, va, hv, cfg)
    return R.hit_ndcg(ranks)[0]


def _cv_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, workers=0):
    import logging
    root = str(tmp_path)
    data = _write_tiny(root)
    monkeypatch.chdir(os.path.join(root, 'work'))
    common = ['--root_data_dir', data] + COMMON_CV + ['--CV_model_load', 'vit-base-patch16-224', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
                                                       '--lr', '1e-3', '--adapter_cv_lr', '1e-3', '--adapter_sasrec_lr', '1e-3', '--label_screen', 'cv']
    common[common.index('--num_workers') + 1] = str(workers)
    a = dict(loss=[], batch=[], eval=[])
    _run_cv(common + ['--epoch', '2'], monkeypatch, a)
    assert a['batch'] == [16, 16, 8] * 2, a['batch']
    assert all(np.isfinite(a['loss'])) and len(a['eval']) >= 3
    ckpts = sorted(os.path.join(dp, f) for dp, _, fs in os.walk('.') for f in fs if f.endswith('.pt'))
    # (the CV entry point saves when the validation HR improves -- run_adapter.py:614-626 --, so epoch 2's file exists only then)
    names = [os.path.basename(c) for c in ckpts]
    assert names[0] == 'epoch-1.pt' and set(names) <= {'epoch-1.pt', 'epoch-2.pt'}, ckpts
    ck = torch.load(ckpts[-1], map_location='cpu', weights_only=False)
    assert set(ck) == {'model_state_dict', 'optimizer', 'rng_state', 'cuda_rng_state'}
    sd = ck['model_state_dict']
    assert any(k.endswith('attention.output.adapter.fc_down.weight') for k in sd) and not any(k.startswith('module.') for k in sd)
    # (b) the HR@10 the run logged at the validation in front of its last save = the oracle's on the weights it saved
    from adapter4rec_amd.cv.data_utils import read_behaviors, read_images
    keys, name2id = read_images(os.path.join(data, 'toy', 'images_log.tsv'))
    item_num, id2keys, tr, va, te, hv, ht = read_behaviors(os.path.join(data, 'toy', 'users_log.tsv'), keys, name2id, 20, 5, logging.getLogger('t'))
    hr = _oracle_hr(sd, data, va, hv, id2keys)
    valids = [h for m, h in a['eval'] if m == 'valid']
    logged = valids[len(names) - 1]
    last_valid = valids[-1]
    print(f'CV: HR@10 logged {logged:.4f} (epoch {len(names)}), oracle on {names[-1]} {hr:.4f}; losses {a["loss"]}')
    assert abs(logged - hr) < 1e-3
    # (a) resume from epoch-1.pt: the uninterrupted run's second epoch
    if len(ckpts) > 1:
        os.remove(ckpts[1])
    b = dict(loss=[], batch=[], eval=[])
    _run_cv(common + ['--epoch', '1', '--load_ckpt_name', 'epoch-1.pt'], monkeypatch, b)
    assert b['batch'] == [16, 16, 8]
    assert abs(b['loss'][0] - a['loss'][3]) < 1e-5 * max(1.0, abs(a['loss'][3])), (b['loss'][0], a['loss'][3])
    np.testing.assert_allclose(b['loss'], a['loss'][3:], rtol=2e-3, atol=2e-3)
    last_b = [h for m, h in b['eval'] if m == 'valid'][-1]
    assert abs(last_b - last_valid) <= 1.0 / N_USERS_T + 1e-9


def _cv_mae_pretrain_then_downstream(tmp_path, monkeypatch):
    """Pretraining/CV (nothing frozen, `--fine_tune_to all --adding_adapter_to None`, CV_model_load mae: Pretraining/CV/script/sm_vit_sasrec.py) saves
    plain-key checkpoints; Downstream/CV loads one by --pretrained_recsys_model (run_adapter.py:341-350), freezes it and trains Houlsby adapters: the
    backbone comes through stage 2 bit for bit, both losses fall."""
    import glob
    import shutil
    root = str(tmp_path)
    data = _write_tiny(root)
    monkeypatch.chdir(os.path.join(root, 'work'))
    base = ['--root_data_dir', data] + COMMON_CV + ['--CV_model_load', 'vit-mae-base']
    a = dict(loss=[], batch=[], eval=[])
    _run_cv(base + ['--adapter_type', 'none', '--adding_adapter_to', 'None', '--fine_tune_to', 'all', '--lr', '1e-3', '--fine_tune_lr', '2e-4',
                    '--label_screen', 'pre', '--epoch', '3'], monkeypatch, a)
    assert all(np.isfinite(a['loss'])) and np.mean(a['loss'][-3:]) < np.mean(a['loss'][:3]), a['loss']
    ck = sorted(glob.glob(os.path.join(root, 'work', 'checkpoint_*', 'cpt_*', 'epoch-*.pt')))[-1:]      # (saved when the validation improves: the latest)
    assert len(ck) == 1, ck
    sd1 = torch.load(ck[0], map_location='cpu', weights_only=False)['model_state_dict']
    assert not any('adapter' in k for k in sd1)
    os.makedirs(os.path.join(root, 'pretrained_models', 'stage1'))
    shutil.copy(ck[0], os.path.join(root, 'pretrained_models', 'stage1', 'epoch-3.pt'))
    b = dict(loss=[], batch=[], eval=[])
    _run_cv(base + ['--adapter_type', 'houslby', '--adding_adapter_to', 'all', '--fine_tune_to', 'None', '--pretrained_recsys_model', 'stage1/epoch-3.pt',
                    '--lr', '1e-3', '--adapter_cv_lr', '1e-3', '--adapter_sasrec_lr', '1e-3', '--label_screen', 'down', '--epoch', '2'], monkeypatch, b)
    assert all(np.isfinite(b['loss'])) and np.mean(b['loss'][-3:]) < np.mean(b['loss'][:3]), b['loss']
    ck2 = [f for f in sorted(glob.glob(os.path.join(root, 'work', 'checkpoint_*', 'cpt_*', 'epoch-*.pt'))) if os.path.dirname(f) != os.path.dirname(ck[0])][-1:]
    assert len(ck2) == 1, ck2
    sd2 = torch.load(ck2[0], map_location='cpu', weights_only=False)['model_state_dict']
    assert any('adapter' in k for k in sd2)
    frozen = [k for k in sd1 if k.endswith('attention.attention.query.weight') or k.endswith('intermediate.dense.weight') or 'patch_embeddings' in k]
    assert len(frozen) >= 5, list(sd1)[:20]
    for k in frozen:
        k2 = k if k in sd2 else k.replace('.attention.output.', '.attention.output.self_output.')
        assert torch.equal(sd1[k], sd2[k2]), k
    assert np.isfinite(b['loss'][0])


@pytest.mark.gpu
def test_cv_run_two_epochs_resume_and_oracle_hr_gpu(tmp_path, monkeypatch):
    _cv_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch)


def test_cv_run_two_epochs_resume_and_oracle_hr_simulated(tmp_path, monkeypatch):
    _simulate_cv(monkeypatch)
    _cv_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch)


def test_cv_run_resume_with_worker_pool_simulated(tmp_path, monkeypatch):
    """with --num_workers > 0 the WORKERS draw the negatives.  They are reseeded per epoch from the torch generator (worker_init_fn
    = the reference's, run_adapter.py:326-334; workers not persistent), so resume = the uninterrupted run holds with a pool as well."""
    _simulate_cv(monkeypatch)
    _cv_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, workers=2)


@pytest.mark.gpu
def test_cv_run_resume_with_worker_pool_gpu(tmp_path, monkeypatch):
    _cv_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, workers=2)


@pytest.mark.gpu
def test_cv_run_mae_pretrain_then_downstream_gpu(tmp_path, monkeypatch):
    _cv_mae_pretrain_then_downstream(tmp_path, monkeypatch)


def test_cv_run_mae_pretrain_then_downstream_simulated(tmp_path, monkeypatch):
    _simulate_cv(monkeypatch)
    _cv_mae_pretrain_then_downstream(tmp_path, monkeypatch)


def _cv_worker_pool_matches_in_process(tmp_path, monkeypatch):
    """--num_workers 2 (records decoded and stacked by DataLoader workers, uploaded / resized / scattered by the training process) draws the same kind
    of batches as --num_workers 0: same batch sizes, finite falling loss, a checkpoint; and image_io.assemble_batch(collate_host(...)) of a sample equals
    the in-process Build_Lmdb_Dataset sample bit for bit when both are fed the same random stream."""
    import random
    from adapter4rec_amd.cv.data_utils import open_image_db, read_behaviors, read_images
    from adapter4rec_amd.cv.image_io import Build_Lmdb_Dataset, assemble_batch, collate_host
    import logging
    root = str(tmp_path)
    data = _write_tiny(root)
    keys, name2id = read_images(os.path.join(data, 'toy', 'images_log.tsv'))
    item_num, id2keys, tr, va, te, hv, ht = read_behaviors(os.path.join(data, 'toy', 'users_log.tsv'), keys, name2id, 20, 5, logging.getLogger('t'))
    db = open_image_db(os.path.join(data, 'toy', 'image.pkl'))
    dev = 'cuda:0' if torch.cuda.is_available() else 'cpu'
    a = Build_Lmdb_Dataset(tr, item_num, 20, db, id2keys, 32, device=dev)
    h = Build_Lmdb_Dataset(tr, item_num, 20, db, id2keys, 32, device=dev, host=True)
    users = sorted(tr)[:3]
    random.seed(5)
    ref = torch.stack([a[u][0] for u in users])
    random.seed(5)
    merged, lm = collate_host([h[u] for u in users])
    got = assemble_batch(merged, 3, 21, 32, torch.device(dev))
    assert torch.equal(ref.cpu(), got.cpu()) and lm.shape == (3, 20)
    monkeypatch.chdir(os.path.join(root, 'work'))
    rec = dict(loss=[], batch=[], eval=[])
    argv = ['--root_data_dir', data] + [x for x in COMMON_CV] + ['--CV_model_load', 'vit-base-patch16-224', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
                                                                 '--lr', '1e-3', '--adapter_cv_lr', '1e-3', '--adapter_sasrec_lr', '1e-3', '--label_screen', 'w2', '--epoch', '2']
    argv[argv.index('--num_workers') + 1] = '2'
    _run_cv(argv, monkeypatch, rec)
    assert rec['batch'] == [16, 16, 8] * 2 and all(np.isfinite(rec['loss'])) and np.mean(rec['loss'][3:]) < np.mean(rec['loss'][:3])


@pytest.mark.gpu
def test_cv_run_worker_pool_gpu(tmp_path, monkeypatch):
    _cv_worker_pool_matches_in_process(tmp_path, monkeypatch)


def test_cv_run_worker_pool_simulated(tmp_path, monkeypatch):
    _simulate_cv(monkeypatch)
    _cv_worker_pool_matches_in_process(tmp_path, monkeypatch)
