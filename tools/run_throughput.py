#!/usr/bin/env python
"""Throughput of the REAL entry point and input path (VERDICT r3 task 3a): adapter4rec_amd/run.py::train on a synthetic
MIND-shaped dataset -- 65 536 news of 28-word titles (30 tokens with [CLS] / [SEP]), users with 23-item histories -- through
read_news_bert / read_behaviors / BuildTrainDataset / DataLoader(num_workers = n, pin_memory) / FlatDDP / FusedAdam, BERT-base geometry
(random init from config.json, as the reference tree ships its pretrained_models/), Houlsby adapters, bf16, B = 32, dropout on: the
configuration bench.py times with device-resident batches.

    python tools/run_throughput.py [--workers 0,4,12] [--users 9600] [--out gpurun_out/run_throughput.json]

Per setting: ONE epoch; the clock runs from the entry of training step 21 to the entry of the last step (torch.cuda.synchronize() at
both ends only), so DataLoader start-up, the first-step buffer allocation and the epoch-end evaluation are outside it, exactly as
bench.py's warm-up is.  Reference: Downstream/Text/run.py:344-357,586-600.
"""
import argparse
import json
import os
import socket
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_ITEMS = 65536


def write_dataset(root, n_users, seed=0):
    rng = np.random.default_rng(seed)
    words = ['w%04d' % i for i in range(4000)]
    d = os.path.join(root, 'pretrained_models', 'bert', 'bert_base_uncased')
    os.makedirs(d)
    with open(os.path.join(d, 'vocab.txt'), 'w') as f:
        f.write('\n'.join(['[PAD]', '[UNK]', '[CLS]', '[SEP]', '[MASK]'] + words) + '\n')
    with open(os.path.join(d, 'config.json'), 'w') as f:        # BERT-base geometry; only the first 4 005 embedding rows are ever addressed
        json.dump(dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                       max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                       attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert'), f)
    data = os.path.join(root, 'data', 'synth')
    os.makedirs(data)
    idx = rng.integers(0, len(words), size=(N_ITEMS, 28))
    with open(os.path.join(data, 'news.tsv'), 'w') as f:
        for i in range(N_ITEMS):
            f.write('N%d\t%s\n' % (i, ' '.join(words[j] for j in idx[i])))
    with open(os.path.join(data, 'behaviors.tsv'), 'w') as f:
        for u in range(n_users):
            seq = rng.choice(N_ITEMS, size=23, replace=False)
            f.write('U%d\t%s\n' % (u, ' '.join('N%d' % i for i in seq)))
    os.makedirs(os.path.join(root, 'work'))
    return os.path.join(root, 'data')


def one_run(data, root, workers, batch, extra):
    import torch.distributed as dist
    from adapter4rec_amd import run
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    stamps = []
    orig = run.FlatDDP.forward

    def fwd(self, *a, **k):
        n = len(stamps)
        if n == 20:                                    # entry of step 21: everything before it is warm-up
            torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        return orig(self, *a, **k)
    run.FlatDDP.forward = fwd
    real_eval = run.run_eval_once
    t_end = []

    def ev(*a, **k):                                   # first thing after the epoch's last step
        if not t_end:
            torch.cuda.synchronize()
            t_end.append(time.perf_counter())
        return real_eval(*a, **k)
    run.run_eval_once = ev
    argv = ['--root_data_dir', data, '--dataset', 'synth', '--behaviors', 'behaviors.tsv', '--news', 'news.tsv', '--mode', 'train',
            '--bert_model_load', 'bert_base_uncased', '--freeze_paras_before', '0', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
            '--fine_tune_to', 'None', '--pretrained_model_name', 'None', '--embedding_dim', '64', '--batch_size', str(batch),
            '--num_workers', str(workers), '--logging_num', '1', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5',
            '--epoch', '1', '--label_screen', 'tp', '--compute_dtype', 'bf16', '--eval_compute_dtype', 'bf16'] + extra
    cwd = os.getcwd()
    os.chdir(os.path.join(root, 'work'))
    try:
        run.main(argv)
    finally:
        os.chdir(cwd)
        run.FlatDDP.forward = orig
        run.run_eval_once = real_eval
        if dist.is_initialized():
            dist.destroy_process_group()
    steps = len(stamps)
    dt = t_end[0] - stamps[20]
    timed = steps - 20
    return dict(num_workers=workers, steps=steps, timed_steps=timed, ms_per_step=round(dt / timed * 1e3, 3),
                user_seq_per_s=round(timed * batch / dt, 1))


def write_cv_dataset(root, n_users, n_items=8192, seed=0):
    """pickled uint8 224 x 224 records (what Build_Lmdb_Dataset decodes: Downstream/CV/data_utils/dataset.py:85-113), ViT-B/16 geometry by default"""
    import pickle
    from adapter4rec_amd.cv.image_io import RecordStore
    rng = np.random.default_rng(seed)
    d = os.path.join(root, 'data', 'synth')
    os.makedirs(d)
    st = RecordStore()
    with open(os.path.join(d, 'images_log.tsv'), 'w') as f:
        for i in range(n_items):
            f.write('v%d\n' % i)
            st.add(('v%d' % i).encode('ascii'), rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), i)
    with open(os.path.join(d, 'image.pkl'), 'wb') as f:
        pickle.dump(dict(st), f)
    with open(os.path.join(d, 'users_log.tsv'), 'w') as f:
        for u in range(n_users):
            seq = rng.choice(n_items, size=23, replace=False)
            f.write('u%d\t%s\n' % (u, ' '.join('v%d' % i for i in seq)))
    os.makedirs(os.path.join(root, 'pretrained_models'), exist_ok=True)
    os.makedirs(os.path.join(root, 'work'), exist_ok=True)
    return os.path.join(root, 'data')


def one_run_cv(data, root, batch, extra, workers=0):
    """adapter4rec_amd/cv/run_adapter.py::train, ViT-B/16 + LoRA r = 8 (bench.py --workload vit_lora's model), one epoch; the clock as in one_run"""
    import torch.distributed as dist
    from adapter4rec_amd.cv import run_adapter as RA
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    stamps, t_end = [], []
    orig, real_eval = RA.FlatDDP.forward, RA.run_eval_once

    def fwd(self, *a, **k):
        if len(stamps) == 20:
            torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        return orig(self, *a, **k)

    def ev(*a, **k):
        if not t_end:
            torch.cuda.synchronize()
            t_end.append(time.perf_counter())
        return 0.0                                      # (the item sweep over 8 192 images is not what is timed here)
    RA.FlatDDP.forward, RA.run_eval_once = fwd, ev
    argv = ['--root_data_dir', data, '--dataset', 'synth', '--lmdb_data', 'image.pkl', '--CV_model_load', 'vit-base-patch16-224', '--CV_resize', '224',
            '--freeze_paras_before', '0', '--adapter_type', 'lora', '--adding_adapter_to', 'all', '--lora_r', '8', '--lora_r_sasrec', '4',
            '--embedding_dim', '64', '--batch_size', str(batch), '--logging_num', '1', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5',
            '--epoch', '1', '--label_screen', 'tp', '--compute_dtype', 'bf16', '--num_workers', str(workers)] + extra
    cwd = os.getcwd()
    os.chdir(os.path.join(root, 'work'))
    try:
        RA.main(argv)
    finally:
        os.chdir(cwd)
        RA.FlatDDP.forward, RA.run_eval_once = orig, real_eval
        if dist.is_initialized():
            dist.destroy_process_group()
    steps = len(stamps)
    dt = t_end[0] - stamps[20]
    timed = steps - 20
    return dict(num_workers=workers, steps=steps, timed_steps=timed, ms_per_step=round(dt / timed * 1e3, 3), user_seq_per_s=round(timed * batch / dt, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cv', action='store_true', help='the image entry point (cv/run_adapter.py::train, ViT-B/16 + LoRA, 8 192 pickled uint8 records)')
    ap.add_argument('--workers', default='0,4,12')
    ap.add_argument('--users', type=int, default=9600)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--out', default='')
    ap.add_argument('--extra', default='', help='extra run.py flags, space separated (e.g. "--device_sampler 1")')
    a = ap.parse_args()
    root = tempfile.mkdtemp(prefix='a4r_tp_')
    t0 = time.time()
    if a.cv:
        batch = a.batch if a.batch != 32 else 8
        data = write_cv_dataset(root, min(a.users, 1600))
        print(f'image dataset written in {time.time() - t0:.1f} s', flush=True)
        r = []
        for w in [int(x) for x in a.workers.split(',')]:
            r.append(one_run_cv(data, root, batch, a.extra.split(), w))
            print(json.dumps(r[-1]), flush=True)
        out = dict(entry_point='adapter4rec_amd/cv/run_adapter.py::train (public path: pickled uint8 records decoded in-process, resized / normalised on the GPU)',
                   batch=batch, runs=r, note='compare with bench.py --workload vit_lora (same model / batch / dtype, device-resident uint8 images)')
        print(json.dumps(out))
        if a.out:
            os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
            with open(a.out, 'w') as f:
                json.dump(out, f, indent=1)
        return
    data = write_dataset(root, a.users)
    print(f'dataset written in {time.time() - t0:.1f} s: {N_ITEMS} news, {a.users} users', flush=True)
    res = []
    for w in [int(x) for x in a.workers.split(',')]:
        r = one_run(data, root, w, a.batch, a.extra.split())
        print(json.dumps(r), flush=True)
        res.append(r)
    out = dict(entry_point='adapter4rec_amd/run.py::train (public path, DataLoader input)', batch=a.batch, runs=res,
               note='compare with bench.py (same model / batch / dtype, device-resident synthetic batches)')
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
