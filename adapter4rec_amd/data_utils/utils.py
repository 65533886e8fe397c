"""Logging / checkpoint helpers with the reference's names (Downstream/Text/data_utils/utils.py)."""
import argparse
import logging
import math
import os
import time

import torch
import torch.distributed as dist


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def setuplogger(dir_label, log_paras, time_run, mode, rank):
    """utils.py:22-56: rank 0 logs INFO to ./logs_<label>_<train|test>/log_<paras><time>.log and stderr, other ranks WARN."""
    code = 'test' if 'test' in mode else 'train'
    fmt = logging.Formatter('[%(levelname)s %(asctime)s] %(message)s')
    Log_file, Log_screen = logging.getLogger('Log_file'), logging.getLogger('Log_screen')
    if rank in (-1, 0):
        path = os.path.join('./logs_' + dir_label + '_' + code)
        os.makedirs(path, exist_ok=True)
        Log_file.setLevel(logging.INFO)
        Log_screen.setLevel(logging.INFO)
        fh = logging.FileHandler(filename=os.path.join(path, 'log_' + log_paras + time_run + '.log'), encoding='utf-8')
        fh.setLevel(logging.INFO)
        fh.setFormatter(fmt)
        sh = logging.StreamHandler()
        sh.setLevel(logging.INFO)
        sh.setFormatter(fmt)
        Log_file.addHandler(fh)
        Log_file.addHandler(sh)
        Log_screen.addHandler(sh)
    else:
        Log_file.setLevel(logging.WARN)
        Log_screen.setLevel(logging.WARN)
    return Log_file, Log_screen


def get_checkpoint(directory, ckpt_name):
    p = os.path.join(directory, ckpt_name)
    return p if os.path.exists(p) else None


def latest_checkpoint(directory, Log_file):
    if not os.path.exists(directory) or not os.listdir(directory):
        return None
    by_epoch = {int(x.split('.')[-2].split('-')[-1]): x for x in os.listdir(directory)}
    return os.path.join(directory, by_epoch[max(by_epoch)]) if by_epoch else None


def get_time(start_time, end_time):
    t = int(end_time - start_time)
    return t // 3600, (t // 60) % 60, t % 60


def para_and_log(model, seq_num, batch_size, Log_file, logging_num, testing_num):
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    Log_file.info('##### total_num {} #####'.format(total))
    Log_file.info('##### trainable_num {} #####'.format(trainable))
    world = dist.get_world_size() if dist.is_initialized() else 1
    steps = math.ceil(seq_num / world / batch_size)
    Log_file.info('##### all {} steps #####'.format(steps))
    per_log, per_test = max(1, int(steps / logging_num)), max(1, int(steps / testing_num))
    Log_file.info('##### {} logs/epoch; {} steps/log #####'.format(logging_num, per_log))
    Log_file.info('##### {} tests/epoch; {} steps/test #####'.format(testing_num, per_test))
    return per_log, per_test


def save_model(now_epoch, model, model_dir, optimizer, rng_state, cuda_rng_state, Log_file):
    """utils.py:109-115: same file name and keys, so checkpoints interchange with the reference."""
    path = os.path.join(model_dir, f'epoch-{now_epoch}.pt')
    m = model.module if hasattr(model, 'module') else model
    torch.save({'model_state_dict': m.state_dict(), 'optimizer': optimizer.state_dict(),
                'rng_state': rng_state, 'cuda_rng_state': cuda_rng_state}, path)
    Log_file.info(f'Model saved to {path}')


def report_time_train(batch_index, now_epoch, loss, next_set_start_time, start_time, Log_file):
    Log_file.info('epoch: {} end, train_loss: {:.5f}'.format(now_epoch, float(loss) / batch_index))
    now = time.time()
    Log_file.info('##### (time) this epoch set: {} hours {} minutes {} seconds #####'.format(*get_time(next_set_start_time, now)))
    Log_file.info('##### (time) start until now: {} hours {} minutes {} seconds #####'.format(*get_time(start_time, now)))
    return time.time()


def report_time_eval(start_time, Log_file):
    Log_file.info('##### (time) eval(valid and test): {} hours {} minutes {} seconds #####'.format(*get_time(start_time, time.time())))
