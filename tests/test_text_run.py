"""End to end through the TEXT entry point (adapter4rec_amd/run.py = Downstream/Text/run.py:288-345,560-646's shape) on a toy
MIND-shaped dataset: tokenizer + config directory -> read_news_bert / read_behaviors -> DataLoader (last batch smaller than the
first) -> FlatDDP -> FusedAdam -> per-epoch valid / test evaluation -> save_model -> RESUME from the epoch-1 checkpoint.

Asserted: (a) the resumed run reproduces the uninterrupted run's second epoch (same per-step losses: weights, Adam moments, the
counter-based dropout stream, the sampler's epoch shuffle and the DataLoader workers' negative sampling all continue from the
checkpoint, as in the reference: run.py:481-492, utils.py:109-115); (b) the HR@10 the run logged equals the CPU oracle's on the
weights it saved; (c) the checkpoint files carry the reference's keys."""
import json
import logging
import os
import socket

import numpy as np
import pytest
import torch

WORDS = ['alpha', 'bravo', 'charlie', 'delta', 'echo', 'foxtrot', 'golf', 'hotel', 'india', 'juliet', 'kilo', 'lima', 'mike',
         'november', 'oscar', 'papa', 'quebec', 'romeo', 'sierra', 'tango', 'uniform', 'victor', 'whiskey', 'xray', 'yankee', 'zulu',
         'red', 'green', 'blue', 'black', 'white', 'north', 'south', 'east', 'west', 'win', 'loss', 'goal', 'vote', 'storm', 'market',
         'court', 'film', 'music', 'health', 'school', 'travel', 'food', 'team', 'city']
N_ITEMS, N_USERS = 60, 40


def write_toy(root):
    """<root>/pretrained_models/bert/bert_tiny/{vocab.txt, config.json}; <root>/data/toy/{news.tsv, behaviors.tsv}; <root>/work (cwd)."""
    rng = np.random.default_rng(0)
    d = os.path.join(root, 'pretrained_models', 'bert', 'bert_tiny')
    os.makedirs(d)
    with open(os.path.join(d, 'vocab.txt'), 'w') as f:
        f.write('\n'.join(['[PAD]', '[UNK]', '[CLS]', '[SEP]', '[MASK]'] + WORDS) + '\n')
    with open(os.path.join(d, 'config.json'), 'w') as f:
        json.dump(dict(vocab_size=5 + len(WORDS), hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                       max_position_embeddings=40, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                       attention_probs_dropout_prob=0.1, pad_token_id=0, model_type='bert'), f)
    data = os.path.join(root, 'data', 'toy')
    os.makedirs(data)
    with open(os.path.join(data, 'news.tsv'), 'w') as f:
        for i in range(N_ITEMS):
            title = ' '.join(rng.choice(WORDS, size=int(rng.integers(3, 14))))
            f.write(f'N{i}\t{title.title()}\n')
    with open(os.path.join(data, 'behaviors.tsv'), 'w') as f:
        for u in range(N_USERS):
            seq = rng.choice(N_ITEMS, size=int(rng.integers(5, 26)), replace=False)      # some longer than max_seq_len + 3: truncated
            f.write(f'U{u}\t' + ' '.join(f'N{i}' for i in seq) + '\n')
    os.makedirs(os.path.join(root, 'work'))
    return os.path.join(root, 'data')


def test_text_host_readers(tmp_path):
    """read_news_bert / read_behaviors / get_doc_input_bert on the toy files (CPU; Downstream/Text/data_utils/preprocess.py:5-151)."""
    from transformers import BertTokenizer
    from adapter4rec_amd.data_utils import get_doc_input_bert, read_behaviors, read_news_bert
    from adapter4rec_amd.parameters import parse_args
    data = write_toy(str(tmp_path))
    tok = BertTokenizer.from_pretrained(str(tmp_path / 'pretrained_models' / 'bert' / 'bert_tiny'))
    args = parse_args(['--num_words_title', '30'])
    id2dic, name2id = read_news_bert(os.path.join(data, 'toy', 'news.tsv'), args, tok)
    item_num, id2dic2, tr, va, te, hv, ht = read_behaviors(os.path.join(data, 'toy', 'behaviors.tsv'), id2dic, name2id, 20, 5, logging.getLogger('t'))
    assert len(tr) == N_USERS and item_num <= N_ITEMS and max(len(s) for s in te.values()) <= 21
    title, mask, *_ = get_doc_input_bert(id2dic2, args)
    assert title.shape == (item_num + 1, 30) and (title[1:, 0] == 2).all() and (mask.sum(1)[1:] >= 5).all()      # [CLS] first, >= 3 words + 2


def test_text_host_readers_news_attributes(tmp_path):
    """abstract / body as the optional 3rd / 4th columns of the news file (the reference's reader names them without ever splitting them off,
    preprocess.py:85-103): the per-attribute [ids | mask] matrices in the order run.py:134-139 concatenates them; a missing column is an error."""
    from transformers import BertTokenizer
    from adapter4rec_amd.data_utils import get_doc_input_bert, read_news_bert
    from adapter4rec_amd.parameters import parse_args
    data = write_toy(str(tmp_path))
    tok = BertTokenizer.from_pretrained(str(tmp_path / 'pretrained_models' / 'bert' / 'bert_tiny'))
    src = os.path.join(data, 'toy', 'news.tsv')
    rows = [l.rstrip('\n').split('\t') for l in open(src)]
    four = os.path.join(data, 'toy', 'news4.tsv')
    with open(four, 'w') as f:
        for name, title in rows:
            f.write(f'{name}\t{title}\t{title} {title}\t{title} ' * 1 + f'{title} {title}\n')
    args = parse_args(['--num_words_title', '30', '--num_words_abstract', '50', '--num_words_body', '40', '--news_attributes', 'title,abstract,body'])
    id2dic, _ = read_news_bert(four, args, tok)
    t, tm, a, am, b, bm = get_doc_input_bert(id2dic, args)
    n = len(rows) + 1
    assert t.shape == (n, 30) and a.shape == am.shape == (n, 50) and b.shape == bm.shape == (n, 40)
    assert (am.sum(1)[1:] >= tm.sum(1)[1:]).all() and (a[1:, 0] == t[1:, 0]).all()
    content = np.concatenate([x for x in (t, tm, a, am, b, bm) if x is not None], axis=1)
    assert content.shape == (n, 2 * (30 + 50 + 40))
    args.news_attributes = ['title', 'abstract']
    with pytest.raises(ValueError):
        read_news_bert(src, args, tok)                          # two columns only


def _run(argv, monkeypatch, record):
    import torch.distributed as dist
    from adapter4rec_amd import run
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    for k, v in dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0').items():
        monkeypatch.setenv(k, v)
    orig_fwd, orig_eval = run.FlatDDP.forward, run.run_eval_once

    def fwd(self, *a, **k):
        out = orig_fwd(self, *a, **k)
        record['loss'].append(out.detach().clone())              # (FlatDDP.forward is the training forward only: eval goes through .module)
        record['batch'].append(int(a[1].shape[0]))
        return out

    def ev(model, item_content, user_history, users_eval, batch_size, item_num, use_modal, mode, local_rank, args, Log_file):
        hit = orig_eval(model, item_content, user_history, users_eval, batch_size, item_num, use_modal, mode, local_rank, args, Log_file)
        record['eval'].append((mode, float(hit)))
        return hit
    monkeypatch.setattr(run.FlatDDP, 'forward', fwd)
    monkeypatch.setattr(run, 'run_eval_once', ev)
    try:
        run.main(argv)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        monkeypatch.setattr(run.FlatDDP, 'forward', orig_fwd)
        monkeypatch.setattr(run, 'run_eval_once', orig_eval)
        for name in ('Log_file', 'Log_screen'):                     # setuplogger adds handlers per call
            lg = logging.getLogger(name)
            for h in list(lg.handlers):
                lg.removeHandler(h)
                h.close()
    record['loss'] = [float(x) for x in record['loss']]


@pytest.mark.gpu
def test_text_run_two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch):
    _two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, [])


def test_text_run_host_logic_simulated(tmp_path, monkeypatch):
    """The same flow on CPU: run.py's host logic (readers, DataLoader, FlatDDP, FusedAdam, eval, save_model, resume) with the kernel
    library replaced by tests/sim_lib.py (a torch restatement of the C ABI; it has no dropout, so the rates are 0 here), gloo instead
    of RCCL and device index 0 mapped to the CPU."""
    _simulate(monkeypatch)
    _two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, ['hidden_dropout_prob', 'attention_probs_dropout_prob'])


def _device_sampler_run(tmp_path, monkeypatch):
    data = write_toy(str(tmp_path))
    cp = os.path.join(str(tmp_path), 'pretrained_models', 'bert', 'bert_tiny', 'config.json')
    c = json.load(open(cp))
    c.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    json.dump(c, open(cp, 'w'))
    monkeypatch.chdir(os.path.join(str(tmp_path), 'work'))
    a = dict(loss=[], batch=[], eval=[])
    _run(['--root_data_dir', data, '--dataset', 'toy', '--behaviors', 'behaviors.tsv', '--news', 'news.tsv', '--mode', 'train',
          '--bert_model_load', 'bert_tiny', '--freeze_paras_before', '0', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
          '--fine_tune_to', 'None', '--pretrained_model_name', 'None', '--embedding_dim', '64', '--batch_size', '16', '--num_workers', '0',
          '--logging_num', '3', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5', '--lr', '1e-3', '--adapter_bert_lr', '1e-3',
          '--adapter_sasrec_lr', '1e-3', '--label_screen', 'dev', '--epoch', '3', '--device_sampler', '1', '--drop_rate', '0',
          '--adapter_dropout_rate', '0'], monkeypatch, a)
    assert a['batch'] == [16, 16, 8] * 3, a['batch']
    assert all(np.isfinite(a['loss'])) and np.mean(a['loss'][-3:]) < np.mean(a['loss'][:3]), a['loss']
    assert len(a['eval']) >= 3


def _pretrain_then_downstream(tmp_path, monkeypatch):
    """The reference's two-stage flow through the one entry point: (1) the Pretraining/ configuration -- no adapters, nothing frozen (`--fine_tune_to
    all --adding_adapter_to None`, Pretraining/Text/script/sm_base_sasrec.py) -- trains every backbone weight and saves epoch checkpoints with PLAIN key
    names; (2) the Downstream/ configuration loads that checkpoint by `--pretrained_model_dir / --pretrained_model_name` (run.py:376-382), freezes it,
    injects Houlsby adapters and trains those only.  Asserted: stage 1's loss falls and its checkpoint holds un-adapted keys; stage 2 starts from
    stage 1's weights (its first evaluation = stage 1's last one: fresh adapters with zero-initialised up-projections ... are NOT zero in the
    reference either, so only 'finite and close'), leaves the backbone untouched and lowers its own loss."""
    import glob
    data = write_toy(str(tmp_path))
    cp = os.path.join(str(tmp_path), 'pretrained_models', 'bert', 'bert_tiny', 'config.json')
    c = json.load(open(cp))
    c.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    json.dump(c, open(cp, 'w'))
    monkeypatch.chdir(os.path.join(str(tmp_path), 'work'))
    common = ['--root_data_dir', data, '--dataset', 'toy', '--behaviors', 'behaviors.tsv', '--news', 'news.tsv', '--mode', 'train',
              '--bert_model_load', 'bert_tiny', '--freeze_paras_before', '0', '--embedding_dim', '64', '--batch_size', '16', '--num_workers', '0',
              '--logging_num', '3', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5', '--drop_rate', '0', '--adapter_dropout_rate', '0']
    a = dict(loss=[], batch=[], eval=[])
    _run(common + ['--adapter_type', 'none', '--adding_adapter_to', 'None', '--fine_tune_to', 'all', '--pretrained_model_name', 'None',
                   '--lr', '1e-3', '--fine_tune_lr', '2e-4', '--label_screen', 'pre', '--epoch', '3'], monkeypatch, a)
    assert all(np.isfinite(a['loss'])) and np.mean(a['loss'][-3:]) < np.mean(a['loss'][:3]), a['loss']
    ck = sorted(glob.glob(os.path.join(str(tmp_path), 'work', 'checkpoint_*', 'cpt_*', 'epoch-3.pt')))
    assert len(ck) == 1, ck
    sd1 = torch.load(ck[0], map_location='cpu')['model_state_dict']
    assert not any('adapter' in k for k in sd1) and any(k.endswith('attention.output.dense.weight') for k in sd1)
    b = dict(loss=[], batch=[], eval=[])
    _run(common + ['--adapter_type', 'houslby', '--adding_adapter_to', 'all', '--fine_tune_to', 'None', '--pretrained_model_dir', os.path.dirname(ck[0]),
                   '--pretrained_model_name', 'epoch-3', '--lr', '1e-3', '--adapter_bert_lr', '1e-3', '--adapter_sasrec_lr', '1e-3',
                   '--label_screen', 'down', '--epoch', '2'], monkeypatch, b)
    assert all(np.isfinite(b['loss'])) and np.mean(b['loss'][-3:]) < np.mean(b['loss'][:3]), b['loss']
    ck2 = [f for f in sorted(glob.glob(os.path.join(str(tmp_path), 'work', 'checkpoint_*', 'cpt_*', 'epoch-2.pt')))
           if os.path.dirname(f) != os.path.dirname(ck[0])]
    assert len(ck2) == 1, ck2
    sd2 = torch.load(ck2[0], map_location='cpu')['model_state_dict']
    assert any('adapter' in k for k in sd2)
    frozen = [k for k in sd1 if k.endswith('attention.self.query.weight') or k.endswith('intermediate.dense.weight') or 'word_embeddings' in k]
    assert len(frozen) >= 5
    for k in frozen:                                     # the pretrained backbone came through stage 2 bit for bit
        assert torch.equal(sd1[k], sd2[k]), k
    assert abs(b['loss'][0] - a['loss'][-1]) < 1.0       # stage 2 starts where stage 1 stopped (fresh adapters perturb it a little)


@pytest.mark.gpu
def test_text_run_pretrain_then_downstream_gpu(tmp_path, monkeypatch):
    _pretrain_then_downstream(tmp_path, monkeypatch)


def test_text_run_pretrain_then_downstream_simulated(tmp_path, monkeypatch):
    _simulate(monkeypatch)
    _pretrain_then_downstream(tmp_path, monkeypatch)


@pytest.mark.gpu
def test_text_run_device_sampler_gpu(tmp_path, monkeypatch):
    """--device_sampler 1 through run.py on the GPU: batches drawn by DeviceTrainSampler on the device (no DataLoader), three epochs of
    16 + 16 + 8 users, finite falling loss, evaluation and checkpoint as usual."""
    _device_sampler_run(tmp_path, monkeypatch)


def test_text_run_device_sampler_simulated(tmp_path, monkeypatch):
    """--device_sampler 1: run.py draws its batches with DeviceTrainSampler instead of BuildTrainDataset + DataLoader -- same epoch
    structure (40 users = 16 + 16 + 8, sharded and shuffled by the DistributedSampler), finite falling loss, evaluation and checkpoint."""
    _simulate(monkeypatch)
    _device_sampler_run(tmp_path, monkeypatch)


def _long_inputs_run(tmp_path, monkeypatch, zero_cfg=False):
    data = write_toy(str(tmp_path))
    if zero_cfg:                                                       # (the simulated long attention has no dropout form)
        cp = os.path.join(str(tmp_path), 'pretrained_models', 'bert', 'bert_tiny', 'config.json')
        c = json.load(open(cp))
        c.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        json.dump(c, open(cp, 'w'))
    monkeypatch.chdir(os.path.join(str(tmp_path), 'work'))
    a = dict(loss=[], batch=[], eval=[])
    _run(['--root_data_dir', data, '--dataset', 'toy', '--behaviors', 'behaviors.tsv', '--news', 'news.tsv', '--mode', 'train',
          '--bert_model_load', 'bert_tiny', '--freeze_paras_before', '0', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
          '--fine_tune_to', 'None', '--pretrained_model_name', 'None', '--embedding_dim', '64', '--batch_size', '16', '--num_workers', '1',
          '--logging_num', '3', '--testing_num', '1', '--max_seq_len', '40', '--num_words_title', '36', '--min_seq_len', '5', '--lr', '1e-3',
          '--adapter_bert_lr', '1e-3', '--adapter_sasrec_lr', '1e-3', '--label_screen', 'dev', '--epoch', '3'], monkeypatch, a)
    assert a['batch'] == [16, 16, 8] * 3, a['batch']
    assert all(np.isfinite(a['loss'])) and np.mean(a['loss'][-3:]) < np.mean(a['loss'][:3]), a['loss']
    assert len(a['eval']) >= 3 and all(0.0 <= h <= 100.0 for _, h in a['eval'])


@pytest.mark.gpu
def test_text_run_long_titles_and_histories_gpu(tmp_path, monkeypatch):
    """run.py with --max_seq_len 40 --num_words_title 36 (both above the short attention kernels' 32): DataLoader batches, three epochs, finite
    falling loss, evaluation (histories of up to 27 ids) and checkpoints -- the long attention kernels in their key-masked / causal forms end to end."""
    _long_inputs_run(tmp_path, monkeypatch)


def test_text_run_long_titles_and_histories_simulated(tmp_path, monkeypatch):
    _simulate(monkeypatch)
    _long_inputs_run(tmp_path, monkeypatch, zero_cfg=True)


def _simulate(monkeypatch):
    import torch.distributed as dist
    import sim_lib
    import adapter4rec_amd.data_utils.metrics as MT
    import adapter4rec_amd.engine as E
    import adapter4rec_amd.optim as O
    from adapter4rec_amd import run
    for mod in (E, O, MT):
        monkeypatch.setattr(mod, 'L', sim_lib)
    monkeypatch.setattr(E.TransRecEngine, '_require_device', lambda self, p0: None)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'get_rng_state', lambda *a: torch.zeros(1, dtype=torch.uint8))
    monkeypatch.setattr(torch.cuda, 'manual_seed_all', lambda s: None)
    real_init = dist.init_process_group
    monkeypatch.setattr(dist, 'init_process_group', lambda backend=None, **k: real_init('gloo', **k))
    on_cpu = lambda a: ['cpu' if (isinstance(x, int) and not isinstance(x, bool)) else x for x in a]
    real_mto, real_tto = torch.nn.Module.to, torch.Tensor.to
    monkeypatch.setattr(torch.nn.Module, 'to', lambda self, *a, **k: real_mto(self, *on_cpu(a), **k))
    monkeypatch.setattr(torch.Tensor, 'to', lambda self, *a, **k: real_tto(self, *on_cpu(a), **{q: v for q, v in k.items() if q != 'non_blocking'}))
    real_parse = run.parse_args
    monkeypatch.setattr(run, 'parse_args', lambda argv=None: real_parse(list(argv) + ['--drop_rate', '0', '--adapter_dropout_rate', '0']))


def _two_epochs_resume_and_oracle_hr(tmp_path, monkeypatch, zero_cfg):
    from oracle import ref_cpu as R
    root = str(tmp_path)
    data = write_toy(root)
    if zero_cfg:
        cp = os.path.join(root, 'pretrained_models', 'bert', 'bert_tiny', 'config.json')
        c = json.load(open(cp))
        c.update({k: 0.0 for k in zero_cfg})
        json.dump(c, open(cp, 'w'))
    monkeypatch.chdir(os.path.join(root, 'work'))                   # load_backbone reads ../pretrained_models/bert/<name> (run.py:289-300)
    common = ['--root_data_dir', data, '--dataset', 'toy', '--behaviors', 'behaviors.tsv', '--news', 'news.tsv', '--mode', 'train',
              '--bert_model_load', 'bert_tiny', '--freeze_paras_before', '0', '--adapter_type', 'houslby', '--adding_adapter_to', 'all',
              '--fine_tune_to', 'None', '--pretrained_model_name', 'None', '--embedding_dim', '64', '--batch_size', '16',
              '--num_workers', '1', '--logging_num', '3', '--testing_num', '1', '--max_seq_len', '20', '--min_seq_len', '5',
              '--lr', '1e-3', '--adapter_bert_lr', '1e-3', '--adapter_sasrec_lr', '1e-3', '--label_screen', 'e2e']
    a = dict(loss=[], batch=[], eval=[])
    _run(common + ['--epoch', '2'], monkeypatch, a)
    assert a['batch'] == [16, 16, 8] * 2, a['batch']                # 40 users, no drop_last: the last batch is smaller (run.py:356)
    assert all(np.isfinite(a['loss'])) and len(a['eval']) >= 3      # valid every epoch (+ test when valid improves / first epoch)
    ckpts = sorted(os.path.join(dp, f) for dp, _, fs in os.walk('.') for f in fs if f.endswith('.pt'))
    assert [os.path.basename(c) for c in ckpts] == ['epoch-1.pt', 'epoch-2.pt'], ckpts
    ck = torch.load(ckpts[1], map_location='cpu', weights_only=False)
    assert set(ck) == {'model_state_dict', 'optimizer', 'rng_state', 'cuda_rng_state'}                    # utils.py:109-115
    sd = ck['model_state_dict']
    assert any(k.endswith('attention.output.adapter.fc_down.weight') for k in sd) and not any(k.startswith('module.') for k in sd)
    st = ck['optimizer']['state']
    assert st and all({'step', 'exp_avg', 'exp_avg_sq'} <= set(v) for v in st.values())                   # torch.optim.Adam's own layout

    # (b) HR@10 the run logged at its last validation == the oracle's on the weights it saved afterwards (epoch-2.pt)
    from transformers import BertTokenizer
    from adapter4rec_amd.data_utils import get_doc_input_bert, read_behaviors, read_news_bert
    from adapter4rec_amd.parameters import parse_args
    args = parse_args(['--num_words_title', '30'])
    tok = BertTokenizer.from_pretrained('../pretrained_models/bert/bert_tiny')
    id2dic, name2id = read_news_bert(os.path.join(data, 'toy', 'news.tsv'), args, tok)
    item_num, id2dic2, tr, va, te, hv, ht = read_behaviors(os.path.join(data, 'toy', 'behaviors.tsv'), id2dic, name2id, 20, 5, logging.getLogger('t'))
    title, mask, *_ = get_doc_input_bert(id2dic2, args)
    content = np.concatenate([title, mask], axis=1).astype(np.int64)
    cfg = dict(R.DEFAULT_CFG, bert_heads=2)
    osd = {k: v.float() for k, v in sd.items()}
    emb = R.item_embeddings(osd, content, cfg)
    _, ranks = R.eval_ranks(osd, emb, va, hv, cfg)
    hr_oracle, _ = R.hit_ndcg(ranks)
    last_valid = [h for m, h in a['eval'] if m == 'valid'][-1]
    print(f'HR@10 logged by run.py at its last validation {last_valid:.4f}, oracle on epoch-2.pt {hr_oracle:.4f}; losses {a["loss"]}')
    assert abs(last_valid - hr_oracle) < 1e-3

    # (a) resume from epoch-1.pt and train ONE more epoch: the uninterrupted run's epoch 2, step for step
    os.remove(ckpts[1])
    b = dict(loss=[], batch=[], eval=[])
    _run(common + ['--epoch', '1', '--load_ckpt_name', 'epoch-1.pt'], monkeypatch, b)
    assert b['batch'] == [16, 16, 8]
    print('uninterrupted epoch 2:', a['loss'][3:], ' resumed:', b['loss'])
    # step 1 after the resume: same weights, same batch (sampler epoch + worker seed from the restored torch RNG state), same dropout
    # masks (engine step counter in the checkpoint); the forward has no atomics except the loss sum itself
    assert abs(b['loss'][0] - a['loss'][3]) < 1e-5 * max(1.0, abs(a['loss'][3])), (b['loss'][0], a['loss'][3])
    # later steps also pass through backward's fp32 atomic accumulation order (not bit-reproducible between two runs)
    np.testing.assert_allclose(b['loss'], a['loss'][3:], rtol=2e-3, atol=2e-3)
    last_b = [h for m, h in b['eval'] if m == 'valid'][-1]
    assert abs(last_b - last_valid) <= 1.0 / N_USERS + 1e-9        # at most one user across the rank-10 boundary
