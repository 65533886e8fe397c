"""Entry point with the reference's shape (Downstream/Text/run.py): parse_args -> init_process_group -> train()/test().

    python -m torch.distributed.run --nproc_per_node N -m adapter4rec_amd.run --mode train --adapter_type houslby ...

Same flags, same epoch loop (zero_grad / forward / backward / step / NaN break / eval / checkpoint), same checkpoint
files.  Differences are the MI355X substitutions only: the model classes come from adapter4rec_amd.model (native
engine), DDP -> FlatDDP (one RCCL all-reduce of the flat adapter-gradient buffer), optim.Adam -> FusedAdam.
"""
import os
import random
import re
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .data_utils import (BuildTrainDataset, eval_model, get_doc_input_bert, get_item_embeddings, read_behaviors,
                         read_news_bert)
from .data_utils.dataset import DeviceTrainSampler
from .data_utils.utils import (get_checkpoint, para_and_log, report_time_eval, report_time_train, save_model, setuplogger)
from .ddp import FlatDDP, any_rank
from .inject import freeze_all, inject_adapters, optimizer_groups
from .model import BertBackbone, Model, ModelCPC
from .optim import FusedAdam
from .parameters import parse_args

DIMS = {'tiny': 128, 'mini': 256, 'medium': 512, 'base': 768, 'large': 1024}


def load_backbone(args, Log_file):
    """run.py:289-300: tokenizer + encoder from ../pretrained_models/{bert,roberta}/<name>.  Weights are loaded with
    HuggingFace when the directory holds them; a config-only directory (as shipped in the reference tree) gives a
    random-init backbone of that geometry."""
    family = 'roberta' if 'roberta' in args.bert_model_load else 'bert'
    path = f'../pretrained_models/{family}/' + args.bert_model_load
    from transformers import BertModel, BertTokenizer, RobertaModel, RobertaTokenizer
    tok = (RobertaTokenizer if family == 'roberta' else BertTokenizer).from_pretrained(path)
    has_weights = any(os.path.exists(os.path.join(path, f)) for f in ('pytorch_model.bin', 'model.safetensors'))
    if has_weights:
        Log_file.info(f'load {family} model...')
        model = (RobertaModel if family == 'roberta' else BertModel).from_pretrained(path, attn_implementation='eager')
    else:
        Log_file.info(f'{path} holds no weights: random-init {family} backbone from config.json')
        model = BertBackbone.from_config_json(os.path.join(path, 'config.json'))
    pooler_para = []
    for key, dim in DIMS.items():
        if key in args.bert_model_load:
            args.word_embedding_dim = dim
            n_layers = {'tiny': 2, 'mini': 4, 'medium': 8, 'base': 12, 'large': 24}[key]
            pooler_para = [5 + 16 * n_layers, 6 + 16 * n_layers]          # run.py:302-316: [37,38] tiny ... [197,198] base
    # run.py:317-319: the first --freeze_paras_before backbone tensors (and the pooler) never train, whatever --fine_tune_to says
    for index, (_, p) in enumerate(model.named_parameters()):
        if index < args.freeze_paras_before or index in pooler_para:
            p.requires_grad = False
    return tok, model


def build_model(args, item_num, use_modal, bert_model, local_rank, Log_file, model_dir):
    model = (ModelCPC if 'cpc' in args.arch else Model)(args, item_num, use_modal, bert_model)
    if 'all' in args.fine_tune_to:
        pass
    elif 'None' in args.fine_tune_to:
        freeze_all(model)
    else:
        raise AssertionError('fine_tune_to should be defined properly')
    if 'None' not in args.pretrained_model_name:                      # pretrained TransRec, plain key names (run.py:376-382)
        ckpt = get_checkpoint(args.pretrained_model_dir, f'{args.pretrained_model_name}.pt')
        if ckpt is None:
            raise FileNotFoundError(f'{args.pretrained_model_dir}/{args.pretrained_model_name}.pt')
        model.load_state_dict(torch.load(ckpt, map_location='cpu')['model_state_dict'])
        Log_file.info(f'Model loaded from {ckpt}')
    model = inject_adapters(model, args)
    start_epoch, ckpt2 = 0, None
    if 'None' not in args.load_ckpt_name:                             # adapter checkpoint, wrapped key names (run.py:481-492)
        path2 = get_checkpoint(model_dir, args.load_ckpt_name)
        ckpt2 = torch.load(path2, map_location='cpu')
        model.load_state_dict(ckpt2['model_state_dict'])
        start_epoch = int(re.split(r'[._-]', args.load_ckpt_name)[1])
        torch.set_rng_state(ckpt2['rng_state'])
        if 'cuda_rng_state' in ckpt2 and torch.cuda.is_available():
            torch.cuda.set_rng_state(ckpt2['cuda_rng_state'])
    if 'None' not in args.adding_adapter_to and 'None' not in args.finetune_layernorm:      # run.py:496-501
        for name, p in model.named_parameters():
            if 'adapter' not in name and ('LayerNorm' in name or 'layer_norm' in name):
                p.requires_grad = True
    return model.to(local_rank), start_epoch, ckpt2


def run_eval_once(model, item_content, user_history, users_eval, batch_size, item_num, use_modal, mode, local_rank, args, Log_file):
    t0 = time.time()
    Log_file.info('Validating...')
    emb = get_item_embeddings(model, item_content, batch_size, args, use_modal, local_rank)
    hit10 = eval_model(model, user_history, users_eval, emb, batch_size, args, item_num, Log_file, mode, local_rank)
    report_time_eval(t0, Log_file)
    return hit10


def train(args, use_modal, local_rank, Log_file, Log_screen, model_dir, start_time):
    tokenizer, bert_model = load_backbone(args, Log_file)
    Log_file.info('read news...')
    before_id2dic, before_name2id = read_news_bert(os.path.join(args.root_data_dir, args.dataset, args.news), args, tokenizer)
    Log_file.info('read behaviors...')
    item_num, id2dic, users_train, users_valid, users_test, hist_valid, hist_test = read_behaviors(
        os.path.join(args.root_data_dir, args.dataset, args.behaviors), before_id2dic, before_name2id,
        args.max_seq_len, args.min_seq_len, Log_file)
    item_content = np.concatenate([x for x in get_doc_input_bert(id2dic, args) if x is not None], axis=1)          # run.py:130-139: [ids | mask] per attribute
    train_dataset = BuildTrainDataset(u2seq=users_train, item_content=item_content, item_num=item_num,
                                      max_seq_len=args.max_seq_len, use_modal=use_modal)
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset)

    def worker_init(worker_id):
        seed = torch.initial_seed() % 2 ** 31 + worker_id + dist.get_rank()
        random.seed(seed)
        np.random.seed(seed)
    train_dl = DataLoader(train_dataset, batch_size=args.batch_size, num_workers=args.num_workers,
                          worker_init_fn=worker_init, pin_memory=True, sampler=sampler)
    model, start_epoch, ckpt2 = build_model(args, item_num, use_modal, bert_model, local_rank, Log_file, model_dir)
    dev_sampler = None
    if getattr(args, 'device_sampler', 0):             # batches drawn on the GPU; the DistributedSampler still shards and shuffles the users
        dev_sampler = DeviceTrainSampler(users_train, item_content, item_num, args.max_seq_len, next(model.parameters()).device,
                                         seed=123456 + dist.get_rank())

        def device_batches():
            ids = list(iter(sampler))
            for i in range(0, len(ids), args.batch_size):
                items, mask = dev_sampler.sample(ids[i:i + args.batch_size])
                yield items.view(-1, args.max_seq_len + 1, 2, items.size(-1)), mask
    model = FlatDDP(model, device_ids=[local_rank], output_device=local_rank)
    optimizer = FusedAdam(optimizer_groups(model, args))
    if ckpt2 is not None:
        optimizer.load_state_dict(ckpt2['optimizer'])
    Log_file.info(model)
    steps_for_log, _ = para_and_log(model, len(users_train), args.batch_size, Log_file, args.logging_num, args.testing_num)
    Log_screen.info('{} train start'.format(args.label_screen))
    next_t = time.time()
    max_eval, max_epoch, max_hit10 = 0, 0, 0
    for ep in range(args.epoch):
        now_epoch = start_epoch + ep + 1
        Log_file.info('epoch {} start'.format(now_epoch))
        loss, batch_index, need_break = 0.0, 1, False
        train_dl.sampler.set_epoch(now_epoch)
        if dev_sampler is not None:
            dev_sampler.set_epoch(now_epoch)
        model.train()
        for sample_items, log_mask in (train_dl if dev_sampler is None else device_batches()):
            # (log_mask stays on the host when it comes from the DataLoader: Model.forward uploads it and the engine reads the batch's pad slots from
            #  the host copy -- short histories' pad items are not encoded; the reference moves both, run.py:591-596)
            sample_items = sample_items.view(-1, sample_items.size(-1))      # (a DataLoader batch stays on the host too: Model.forward reads its longest title, then uploads it)
            if dev_sampler is not None:
                sample_items = sample_items.to(local_rank, non_blocking=True)
            optimizer.zero_grad()
            bz_loss = model(sample_items, log_mask, local_rank)
            loss += bz_loss.detach()
            bz_loss.backward()
            optimizer.step()
            if batch_index % steps_for_log == 0:                      # the only host sync: NaN check + log line
                if any_rank(torch.isnan(loss)):                          # every rank breaks together: the evaluation below uses collectives
                    need_break = True
                    break
                Log_file.info('cnt: {}, Ed: {}, batch loss: {:.5f}, sum loss: {:.5f}'.format(
                    batch_index, batch_index * args.batch_size, loss.item() / batch_index, loss.item()))
            batch_index += 1
        if not need_break and any_rank(torch.isnan(loss)):                 # a NaN after the last log step of the epoch: the reference checks
            need_break = True                                          # every batch (run.py:601-603); never evaluate / save NaN weights
        if not need_break:
            hit10 = run_eval_once(model, item_content, hist_valid, users_valid, 512, item_num, use_modal, 'valid', local_rank, args, Log_file)
            if hit10 > max_eval:
                max_eval, max_epoch = hit10, now_epoch
            model.train()
            if max_eval > max_hit10 or max_hit10 == 0 or ep % 10 == 0:
                max_hit10 = max(max_hit10, max_eval)
                run_eval_once(model, item_content, hist_test, users_test, 512, item_num, use_modal, 'test', local_rank, args, Log_file)
                if use_modal and dist.get_rank() == 0:
                    save_model(now_epoch, model, model_dir, optimizer, torch.get_rng_state(), torch.cuda.get_rng_state(), Log_file)
        next_t = report_time_train(batch_index, now_epoch, loss, next_t, start_time, Log_file)
        Log_screen.info('{} training: epoch {}/{}'.format(args.label_screen, now_epoch, args.epoch))
        if need_break:
            break
    if dist.get_rank() == 0:
        save_model(now_epoch, model, model_dir, optimizer, torch.get_rng_state(), torch.cuda.get_rng_state(), Log_file)
    Log_file.info(' max eval Hit10 {:0.5f}  in epoch {}'.format(max_eval * 100, max_epoch))


def test(args, use_modal, local_rank, Log_file, Log_screen, model_dir, start_time):
    tokenizer, bert_model = load_backbone(args, Log_file)
    before_id2dic, before_name2id = read_news_bert(os.path.join(args.root_data_dir, args.dataset, args.news), args, tokenizer)
    item_num, id2dic, _, users_valid, users_test, hist_valid, hist_test = read_behaviors(
        os.path.join(args.root_data_dir, args.dataset, args.behaviors), before_id2dic, before_name2id,
        args.max_seq_len, args.min_seq_len, Log_file)
    item_content = np.concatenate([x for x in get_doc_input_bert(id2dic, args) if x is not None], axis=1)
    model, _, _ = build_model(args, item_num, use_modal, bert_model, local_rank, Log_file, model_dir)
    model = FlatDDP(model, device_ids=[local_rank], output_device=local_rank)
    run_eval_once(model, item_content, hist_valid, users_valid, 512, item_num, use_modal, 'valid', local_rank, args, Log_file)
    run_eval_once(model, item_content, hist_test, users_test, 512, item_num, use_modal, 'test', local_rank, args, Log_file)


def setup_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def main(argv=None):
    args = parse_args(argv)
    local_rank = args.local_rank if args.local_rank >= 0 else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend='nccl', init_method='env://')      # 'nccl' is RCCL on ROCm
    setup_seed(123456)
    use_modal = True                                                   # run.py:687
    dir_label = str(args.arch) + f'_{args.bert_model_load}_freeze_{args.freeze_paras_before}' + f'_{args.adapter_type}'
    log_paras = f'bs_{args.batch_size}_ed_{args.embedding_dim}_lr_{args.lr}_Flr_{args.fine_tune_lr}_dtype_{args.compute_dtype}'
    model_dir = os.path.join('./checkpoint_' + dir_label, 'cpt_' + log_paras)
    time_run = time.strftime('-%Y%m%d-%H%M%S', time.localtime())
    args.label_screen = args.label_screen + time_run
    Log_file, Log_screen = setuplogger(dir_label, log_paras, time_run, args.mode, dist.get_rank())
    Log_file.info(args)
    os.makedirs(model_dir, exist_ok=True)
    t0 = time.time()
    (test if 'test' in args.mode else train)(args, use_modal, local_rank, Log_file, Log_screen, model_dir, t0)
    hours, minutes, seconds = (lambda t: (t // 3600, (t // 60) % 60, t % 60))(int(time.time() - t0))
    Log_file.info('##### (time) all: {} hours {} minutes {} seconds #####'.format(hours, minutes, seconds))


if __name__ == '__main__':
    main()
