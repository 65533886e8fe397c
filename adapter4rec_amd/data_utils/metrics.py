"""HR@10 / nDCG@10 evaluation with the reference's entry points (Downstream/Text/data_utils/metrics.py:62-116).

Same contract -- ``get_item_embeddings`` encodes all N+1 items with ``model.module.bert_encoder``; ``eval_model`` shards
users like ``SequentialDistributedSampler``, takes the last position of ``user_encoder``, scores against all items, masks
the history, drops the pad column and reports mean Hit@10 / nDCG@10 -- but the per-user Python loop, the [bs, N] fp64
label matrix and the argsort are replaced by one launch of a4r_eval_rank (rank of the target, history excluded, no
[users, items] matrix in HBM)."""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib as L
from .dataset import SequentialDistributedSampler


def print_metrics(x, Log_file, v_or_t):
    Log_file.info(v_or_t + '_results   {}'.format('\t'.join(['{:0.5f}'.format(i * 100) for i in x])))


def _inner(model, args):
    m = model.module if hasattr(model, 'module') else model
    return m.model if 'compacter' in args.adapter_type and hasattr(m, 'model') else m


def get_item_embeddings(model, item_content, test_batch_size, args, use_modal, local_rank):
    """metrics.py:62-79 -> fp32 [N + 1, E] (on the device; the reference returns it on the CPU and moves it back)."""
    model.eval()
    inner = _inner(model, args)
    enc = inner.bert_encoder
    ed = getattr(args, 'eval_compute_dtype', None)
    if ed and ed != inner.compute_dtype:       # e.g. the item sweep in fp32 (the reference's eval precision) under bf16 training
        enc = inner.item_encoder_in(ed)
    dev = next(model.parameters()).device
    content = torch.as_tensor(np.asarray(item_content)).long()
    lo, hi, chunk, world = _my_shard(content.shape[0])
    out = []
    # the reference's test_batch_size (512 at run.py:650) is sized for ITS activation memory; the native encoder takes 4 096 titles per
    # call (fp32 sweep of 65 537 titles: 3.44 s at 512, 2.63 s = 84 % of the exact-fp32 MFMA peak at 4 096; profiles/r03_a_eval_bench.json)
    # A4R_EVAL_SWEEP_BATCH overrides it either way (e.g. a smaller sweep to fit memory next to a large training state).
    step = int(os.environ.get('A4R_EVAL_SWEEP_BATCH', 0)) or max(int(test_batch_size), 4096)
    with torch.no_grad():
        for i in range(lo, hi, step):
            out.append(enc(content[i:min(i + step, hi)].to(dev)))
    return _gather_shards(torch.cat(out, 0) if out else torch.zeros(0, enc_dim(model, args), device=dev), content.shape[0], chunk, world)


def enc_dim(model, args):
    return int(args.embedding_dim)


def _my_shard(n):
    """The item sweep is sharded over the data-parallel ranks (the reference encodes all N + 1 items on EVERY rank,
    metrics.py:62-79): rank r takes rows [r * chunk, (r + 1) * chunk) and the table is all-gathered."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    r = dist.get_rank() if dist.is_initialized() else 0
    chunk = (n + world - 1) // world
    return min(r * chunk, n), min((r + 1) * chunk, n), chunk, world


def _gather_shards(mine, n, chunk, world):
    if world == 1:
        return mine
    pad = torch.zeros(chunk, mine.shape[1], dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat(parts, 0)[:n].contiguous()


def eval_ranks(model, user_history, eval_seq, item_embeddings, test_batch_size, args, user_ids):
    """Rank (1 = best) of each listed user's held-out target among all items not in the user's history."""
    inner = _inner(model, args)
    dev = next(model.parameters()).device
    emb = item_embeddings.to(dev).float().contiguous()
    Lm = args.max_seq_len + 1
    E = emb.shape[1]
    ranks = []
    with torch.no_grad():
        for s in range(0, len(user_ids), test_batch_size):
            users = user_ids[s:s + test_batch_size]
            ids = np.zeros((len(users), Lm - 1), dtype=np.int64)
            mask = np.zeros((len(users), Lm - 1), dtype=np.float32)
            target, ptr, hist = [], [0], []
            for r, u in enumerate(users):
                seq = list(eval_seq[u])
                toks = seq[:-1]
                pad = Lm - len(seq)
                ids[r, pad:] = toks
                mask[r, pad:] = 1.0
                target.append(seq[-1])
                h = [int(x) for x in np.asarray(user_history[u]).reshape(-1)]
                if len(h) > L.EVAL_MAX_HISTORY:        # never truncate: an unmasked history item changes ranks silently
                    raise ValueError(f'user {u}: history of {len(h)} items exceeds A4R_EVAL_MAX_HISTORY = {L.EVAL_MAX_HISTORY} '
                                     f'(the reference keeps max_seq_len + 2 = {args.max_seq_len + 2})')
                hist += h
                ptr.append(len(hist))
            ids_t = torch.from_numpy(ids).to(dev)
            input_embs = emb[ids_t.view(-1)].view(len(users), Lm - 1, E)
            prec = inner.user_encoder(input_embs, torch.from_numpy(mask).to(dev), None)[:, -1].contiguous()
            rank = torch.zeros(len(users), dtype=torch.int32, device=dev)
            L.eval_rank(prec, emb, torch.tensor(target, dtype=torch.int32, device=dev),
                        torch.tensor(ptr, dtype=torch.int32, device=dev),
                        torch.tensor(hist + [0], dtype=torch.int32, device=dev), rank)
            ranks.append(rank)
    return torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int32, device=dev)


def eval_model(model, user_history, eval_seq, item_embeddings, test_batch_size, args, item_num, Log_file, v_or_t, local_rank):
    """metrics.py:82-116.  Returns mean Hit@10 over all users (ranks are gathered over the data-parallel group)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank_id = dist.get_rank() if dist.is_initialized() else 0
    n_users = len(eval_seq)
    sampler = SequentialDistributedSampler(list(range(n_users)), test_batch_size, rank=rank_id, num_replicas=world)
    user_ids = list(iter(sampler))
    model.eval()
    topK = 10
    Log_file.info(v_or_t + '_methods   {}'.format('\t'.join(['Hit{}'.format(topK), 'nDCG{}'.format(topK)])))
    ranks = eval_ranks(model, user_history, eval_seq, item_embeddings, test_batch_size, args, user_ids).float()
    hit = (ranks <= topK).float()
    ndcg = torch.where(ranks <= topK, 1.0 / torch.log2(ranks + 1.0), torch.zeros_like(ranks))
    if world > 1:
        parts_h = [torch.zeros_like(hit) for _ in range(world)]
        parts_n = [torch.zeros_like(ndcg) for _ in range(world)]
        dist.all_gather(parts_h, hit)
        dist.all_gather(parts_n, ndcg)
        hit, ndcg = torch.cat(parts_h), torch.cat(parts_n)
    mean_eval = [float(hit[:n_users].mean().item()), float(ndcg[:n_users].mean().item())]
    print_metrics(mean_eval, Log_file, v_or_t)
    return mean_eval[0]
