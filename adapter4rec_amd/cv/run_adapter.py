"""Entry point with the reference's shape (Downstream/CV/run_adapter.py): parse_args -> init_process_group -> train()/test().

    python -m torch.distributed.run --nproc_per_node N -m adapter4rec_amd.cv.run_adapter --CV_model_load vit-base-patch16-224 \\
        --adapter_type houslby --adding_adapter_to all --max_seq_len 20 --batch_size 8 ...

Same flags and epoch loop; the MI355X substitutions: model classes from adapter4rec_amd.cv (native engine), DDP -> FlatDDP,
optim.Adam -> FusedAdam, the CPU image transform -> uint8 records resized and normalised on the GPU (cv/image_io.py), fp16
autocast + GradScaler (--use_scale half) -> bf16 storage with fp32 accumulation (no loss scaling needed)."""
import os
import random
import re
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from ..data_utils.utils import get_checkpoint, para_and_log, report_time_eval, report_time_train, save_model, setuplogger
from ..ddp import FlatDDP, any_rank
from ..inject import freeze_all
from ..optim import FusedAdam
from . import Model, ModelCPC, ViTForImageClassification, ViTMAEModel
from .data_utils import eval_model, get_itemLMDB_embeddings, open_image_db, read_behaviors, read_images
from .image_io import Build_Lmdb_Dataset, assemble_batch, collate_host
from .inject import inject_adapters, optimizer_groups
from .parameters import parse_args


def load_backbone(args, Log_file):
    """run_adapter.py:286-297.  HuggingFace weights are mapped onto the 4.20.1-shaped containers when the directory holds them
    (transformers >= 5 renamed the modules); a config-only directory gives a random-init backbone of that geometry."""
    mae = 'mae' in args.CV_model_load
    path = '../pretrained_models/' + (args.CV_model_load if mae else 'vit-base-patch16-224')
    geom = None
    if os.path.exists(os.path.join(path, 'config.json')):              # (HF from_pretrained reads the geometry from the directory's config.json)
        import json
        with open(os.path.join(path, 'config.json')) as f:
            geom = {k: v for k, v in json.load(f).items() if not isinstance(v, (dict, list))}
    net = ViTMAEModel(geom) if mae else ViTForImageClassification(geom)
    weights = [os.path.join(path, f) for f in ('model.safetensors', 'pytorch_model.bin') if os.path.exists(os.path.join(path, f))]
    if weights:
        Log_file.info(f'load {path} ...')
        if weights[0].endswith('.safetensors'):
            from safetensors.torch import load_file
            sd = load_file(weights[0])
        else:
            sd = torch.load(weights[0], map_location='cpu')
        if mae:
            sd = {k[len('vit.'):] if k.startswith('vit.') else k: v for k, v in sd.items() if not k.startswith('decoder')}
        missing = net.load_state_dict({k: v for k, v in sd.items() if not k.startswith('classifier') and 'pooler' not in k}, strict=False)
        Log_file.info(f'missing keys: {[k for k in missing.missing_keys if "classifier" not in k]}')
    else:
        Log_file.info(f'{path} holds no weights: random-init backbone')
    if not mae:                                                        # run_adapter.py:291-296
        net.classifier = torch.nn.Linear(net.config['hidden_size'], args.embedding_dim)
        torch.nn.init.xavier_normal_(net.classifier.weight.data)
        torch.nn.init.constant_(net.classifier.bias.data, 0)
    for index, (_, p) in enumerate(net.named_parameters()):            # :299-301
        if index < args.freeze_paras_before:
            p.requires_grad = False
    return net


def build_model(args, item_num, use_modal, cv_model, local_rank, Log_file, model_dir):
    model = (ModelCPC if 'cpc' in args.arch else Model)(args, item_num, use_modal, cv_model)
    if 'None' not in args.pretrained_recsys_model:                    # :341-350
        ckpt = torch.load(get_checkpoint('../pretrained_models/', args.pretrained_recsys_model), map_location='cpu')
        model.load_state_dict(ckpt['model_state_dict'])
    if 'all' in args.fine_tune_to:                                    # Downstream/CV/run.py: end-to-end fine-tuning (ViT, not ViT-MAE)
        pass
    elif 'None' in args.fine_tune_to:
        freeze_all(model)
    else:
        raise AssertionError('fine_tune_to should be defined properly')
    model = inject_adapters(model, args)
    start_epoch, ckpt2 = 0, None
    if 'None' not in args.load_ckpt_name:
        ckpt2 = torch.load(get_checkpoint(model_dir, args.load_ckpt_name), map_location='cpu')
        model.load_state_dict(ckpt2['model_state_dict'])
        start_epoch = int(re.split(r'[._-]', args.load_ckpt_name)[1])
        torch.set_rng_state(ckpt2['rng_state'])
        if 'cuda_rng_state' in ckpt2 and torch.cuda.is_available():
            torch.cuda.set_rng_state(ckpt2['cuda_rng_state'])
    if 'None' not in args.finetune_layernorm:                         # :484-488
        for name, p in model.named_parameters():
            if 'adapter' not in name and ('LayerNorm' in name or 'layer_norm' in name or 'layernorm' in name):
                p.requires_grad = True
    return model.to(local_rank), start_epoch, ckpt2


def run_eval_once(model, db, item_id_to_keys, user_history, users_eval, batch_size, item_num, mode, local_rank, args, Log_file):
    t0 = time.time()
    Log_file.info('Validating...')
    emb = get_itemLMDB_embeddings(model, item_num, item_id_to_keys, batch_size, args, local_rank, db=db)
    hit10 = eval_model(model, user_history, users_eval, emb, batch_size, args, item_num, Log_file, mode, local_rank)
    report_time_eval(t0, Log_file)
    return hit10


def _collate(batch):
    return torch.stack([b[0] for b in batch]), torch.stack([b[1] for b in batch])


def train(args, use_modal, local_rank, Log_file, Log_screen, model_dir, start_time):
    cv_model = load_backbone(args, Log_file)
    before_keys, before_name2id = read_images(os.path.join(args.root_data_dir, args.dataset, args.images))
    item_num, item_id_to_keys, users_train, users_valid, users_test, hist_valid, hist_test = read_behaviors(
        os.path.join(args.root_data_dir, args.dataset, args.behaviors), before_keys, before_name2id, args.max_seq_len,
        args.min_seq_len, Log_file)
    db = open_image_db(os.path.join(args.root_data_dir, args.dataset, args.lmdb_data))
    # --num_workers n > 0 (the reference's DataLoader pool, run_adapter.py:448-450; its 12 workers do PIL resizing on the CPU): the workers decode the
    # records and stack them by source size (no device in a worker), the pinned batches are uploaded, resized and scattered on the GPU here
    # (image_io.assemble_batch) while the previous step computes.  --num_workers 0: decode in this process.
    host = args.num_workers > 0
    train_dataset = Build_Lmdb_Dataset(users_train, item_num, args.max_seq_len, db, item_id_to_keys, args.CV_resize, device=f'cuda:{local_rank}', host=host)
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset)
    if host:
        # The workers draw the negatives (Python `random`, Build_Lmdb_Dataset.__getitem__): reseeded as the reference does (run_adapter.py:326-334)
        # from the worker's torch seed + worker id + rank.  NOT persistent: every epoch's iterator takes a fresh base seed from the torch generator
        # -- the state a checkpoint holds (utils.py:109-115) --, so a resumed run's workers draw what the uninterrupted run's would have, and two
        # ranks never share a negative stream.
        def worker_init(worker_id):
            seed = torch.initial_seed() % 2 ** 31 + worker_id + dist.get_rank()
            random.seed(seed)
            np.random.seed(seed)
        train_dl = DataLoader(train_dataset, batch_size=args.batch_size, num_workers=args.num_workers, sampler=sampler, collate_fn=collate_host,
                              worker_init_fn=worker_init, pin_memory=True, prefetch_factor=2)
    else:
        train_dl = DataLoader(train_dataset, batch_size=args.batch_size, num_workers=0, sampler=sampler, collate_fn=_collate)
    model, start_epoch, ckpt2 = build_model(args, item_num, use_modal, cv_model, local_rank, Log_file, model_dir)
    model = FlatDDP(model, device_ids=[local_rank], output_device=local_rank)
    optimizer = FusedAdam(optimizer_groups(model, args))
    if ckpt2 is not None:
        optimizer.load_state_dict(ckpt2['optimizer'])
    steps_for_log, _ = para_and_log(model, len(users_train), args.batch_size, Log_file, args.logging_num, args.testing_num)
    Log_screen.info('{} train start'.format(args.label_screen))
    next_t = time.time()
    max_eval, max_epoch, max_hit10, now_epoch = 0, 0, 0, start_epoch
    R = args.CV_resize
    for ep in range(args.epoch):
        now_epoch = start_epoch + ep + 1
        Log_file.info('epoch {} start'.format(now_epoch))
        loss, batch_index, need_break = 0.0, 1, False
        model.train()
        train_dl.sampler.set_epoch(now_epoch)
        # --num_workers 0: Build_Lmdb_Dataset draws its negatives from Python's `random` in THIS process, and the checkpoint holds the torch RNG state
        # only (utils.py:109-115): the epoch's stream is re-seeded from the torch generator, so a resumed run draws the negatives the uninterrupted
        # run would have.  With a worker pool the draw below is kept (both runs consume the same generator state) and the workers' streams come from
        # the iterator's base seed (worker_init above) -- the text entry point's arrangement.
        random.seed(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        for sample_items, log_mask in train_dl:
            if host:
                sample_items = assemble_batch(sample_items, log_mask.shape[0], args.max_seq_len + 1, R, torch.device('cuda', local_rank))
            sample_items = sample_items.view(-1, R, R, 3)              # uint8 HWC (the reference: .view(-1, 3, R, R) of fp32)
            optimizer.zero_grad()                                      # (log_mask stays on the host: Model.forward uploads it, the engine reads the pad slots from it)
            bz_loss = model(sample_items, log_mask, local_rank)
            loss += bz_loss.detach()
            bz_loss.backward()
            optimizer.step()
            if batch_index % steps_for_log == 0:
                if any_rank(torch.isnan(loss)):                          # every rank breaks together: the evaluation below uses collectives
                    need_break = True
                    break
                Log_file.info('cnt: {}, Ed: {}, batch loss: {:.5f}, sum loss: {:.5f}'.format(
                    batch_index, batch_index * args.batch_size, loss.item() / batch_index, loss.item()))
            batch_index += 1
        if not need_break and any_rank(torch.isnan(loss)):                 # a NaN after the last log step: never evaluate / save NaN weights
            need_break = True
        if not need_break:
            hit10 = run_eval_once(model, db, item_id_to_keys, hist_valid, users_valid, 256, item_num, 'valid', local_rank, args, Log_file)
            if hit10 > max_eval:
                max_eval, max_epoch = hit10, now_epoch
            if max_eval > max_hit10 or max_hit10 == 0 or ep % 10 == 0:
                max_hit10 = max(max_hit10, max_eval)
                run_eval_once(model, db, item_id_to_keys, hist_test, users_test, 256, item_num, 'test', local_rank, args, Log_file)
                if dist.get_rank() == 0:
                    save_model(now_epoch, model, model_dir, optimizer, torch.get_rng_state(), torch.cuda.get_rng_state(), Log_file)
        next_t = report_time_train(batch_index, now_epoch, loss, next_t, start_time, Log_file)
        if need_break:
            break
    Log_file.info(' max eval Hit10 {:0.5f}  in epoch {}'.format(max_eval * 100, max_epoch))


def test(args, use_modal, local_rank, Log_file, Log_screen, model_dir, start_time):
    cv_model = load_backbone(args, Log_file)
    before_keys, before_name2id = read_images(os.path.join(args.root_data_dir, args.dataset, args.images))
    item_num, item_id_to_keys, _, users_valid, users_test, hist_valid, hist_test = read_behaviors(
        os.path.join(args.root_data_dir, args.dataset, args.behaviors), before_keys, before_name2id, args.max_seq_len,
        args.min_seq_len, Log_file)
    db = open_image_db(os.path.join(args.root_data_dir, args.dataset, args.lmdb_data))
    model, _, _ = build_model(args, item_num, use_modal, cv_model, local_rank, Log_file, model_dir)
    model = FlatDDP(model, device_ids=[local_rank], output_device=local_rank)
    run_eval_once(model, db, item_id_to_keys, hist_valid, users_valid, 256, item_num, 'valid', local_rank, args, Log_file)
    run_eval_once(model, db, item_id_to_keys, hist_test, users_test, 256, item_num, 'test', local_rank, args, Log_file)


def main(argv=None):
    args = parse_args(argv)
    local_rank = args.local_rank if args.local_rank >= 0 else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend='nccl', init_method='env://')      # 'nccl' is RCCL on ROCm
    torch.manual_seed(123456)
    np.random.seed(123456)
    random.seed(123456)
    use_modal = 'modal' in args.item_tower
    dir_label = f'{args.arch}_{args.CV_model_load}_freeze_{args.freeze_paras_before}_{args.adapter_type}'
    log_paras = f'bs_{args.batch_size}_ed_{args.embedding_dim}_lr_{args.lr}_Flr_{args.fine_tune_lr}_dtype_{args.compute_dtype}'
    model_dir = os.path.join('./checkpoint_' + dir_label, 'cpt_' + log_paras)
    time_run = time.strftime('-%Y%m%d-%H%M%S', time.localtime())
    args.label_screen = args.label_screen + time_run
    Log_file, Log_screen = setuplogger(dir_label, log_paras, time_run, args.mode, dist.get_rank())
    Log_file.info(args)
    os.makedirs(model_dir, exist_ok=True)
    t0 = time.time()
    (test if 'test' in args.mode else train)(args, use_modal, local_rank, Log_file, Log_screen, model_dir, t0)


if __name__ == '__main__':
    main()
