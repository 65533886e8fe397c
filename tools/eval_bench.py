#!/usr/bin/env python
"""Throughput of the native evaluation (reference: Downstream/Text/data_utils/metrics.py:62-116).

  python tools/eval_bench.py [--items 65536] [--users 32768] [--json out.json]

Times, on one MI355X, with BERT-base + Houlsby weights (random init, as bench.py):
  * the item sweep `get_item_embeddings` (all N + 1 titles through the item encoder, forward only) in the eval dtypes
    (--eval_compute_dtype fp32 = the default: exact-fp32 MFMA; bf16), at run.py's batch of 512 titles and at 4 096;
  * `eval_model`: user encoder + a4r_eval_rank over `users` users at run.py's batch of 512 users, and the rank kernel alone
    (HIP events) against its HBM/L2 roofline: the kernel streams the N x E fp32 item table once per 16 users.
Prints one JSON object; the numbers land in profiles/r03_*_eval_bench.json."""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def timed(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--items', type=int, default=65536)
    ap.add_argument('--users', type=int, default=32768)
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from adapter4rec_amd import _lib as L
    from adapter4rec_amd.data_utils import eval_model, get_item_embeddings
    args = B.make_args(32, 'bf16')
    model, _ = B.build_model(args, dev)
    model.eval()
    g = torch.Generator().manual_seed(B.SEED)
    content = B.synth_content(a.items, g).numpy()
    res = dict(items=a.items + 1, tokens_per_item=30, encoder='BERT-base + Houlsby (random init)', sweep={})
    # forward FLOPs per item: 12 layers x 30 tokens x (2 x 14 155 776 dense + attention 4 S H) + adapters 2 x 2 x 2 x 768 x 64
    flop_item = 12 * 30 * (2 * 7077888 + 4 * 30 * 768 + 2 * 2 * 2 * 768 * 64)
    emb = None
    for dt in ('fp32', 'bf16'):
        for bs in (512, 4096):
            args.eval_compute_dtype = dt
            get_item_embeddings(model, content[:bs * 2], bs, args, True, 0)             # warm-up: snapshot engine build + buffers
            t, e = timed(lambda: get_item_embeddings(model, content, bs, args, True, 0))
            peak = 157.3 if dt == 'fp32' else 2500.0
            res['sweep'][f'{dt}_batch{bs}'] = dict(seconds=round(t, 3), items_per_s=round((a.items + 1) / t, 1),
                                                   tflops=round((a.items + 1) * flop_item / t / 1e12, 1),
                                                   frac_of_mfma_peak=round((a.items + 1) * flop_item / t / 1e12 / peak, 3), peak_tflops=peak)
            if dt == 'fp32' and bs == 512:
                emb = e
    # users: random histories of 5..22 items, the last one held out
    rng = np.random.default_rng(B.SEED)
    eval_seq, hist = {}, {}
    for u in range(a.users):
        n = int(rng.integers(5, 22))                                # <= max_seq_len + 1 = 21 (data_utils/preprocess.py:48-59)
        seq = [int(x) for x in rng.integers(1, a.items + 1, size=n)]
        eval_seq[u], hist[u] = seq, torch.LongTensor(seq[:-1])
    log = logging.getLogger('eval-bench')
    eval_model(model, {u: hist[u] for u in range(1024)}, {u: eval_seq[u] for u in range(1024)}, emb, 512, args, a.items, log, 'valid', 0)
    t, hr = timed(lambda: eval_model(model, hist, eval_seq, emb, 512, args, a.items, log, 'valid', 0))
    res['eval_model'] = dict(users=a.users, batch=512, seconds=round(t, 3), users_per_s=round(a.users / t, 1), hr10=hr,
                             note='first evaluation of these users: includes building the evaluation set (ids, masks, history CSR) from the two dicts')
    t2, hr2 = timed(lambda: eval_model(model, hist, eval_seq, emb, 512, args, a.items, log, 'valid', 0), reps=3)
    res['eval_model_repeat'] = dict(users=a.users, seconds=round(t2, 4), users_per_s=round(a.users / t2, 1), hr10=hr2,
                                    note='every later evaluation of a run (run.py evaluates the same users each epoch: the set is cached on the device)')
    # the rank kernel alone
    E = emb.shape[1]
    for U in (512, 4096, 32768):
        prec = torch.randn(U, E, device=dev)
        target = torch.randint(1, a.items + 1, (U,), device=dev, dtype=torch.int32)
        ptr = torch.arange(0, U + 1, device=dev, dtype=torch.int32) * 20
        hidx = torch.randint(1, a.items + 1, (U * 20 + 1,), device=dev, dtype=torch.int32)
        rank = torch.zeros(U, dtype=torch.int32, device=dev)
        for _ in range(2):
            L.eval_rank(prec, emb, target, ptr, hidx, rank)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.eval_rank(prec, emb, target, ptr, hidx, rank)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        table_bytes = (a.items + 1) * E * 4
        streamed = table_bytes * ((U + 15) // 16)                   # algorithmic: the table once per 16 users (from L2 / Infinity Cache / HBM)
        res[f'eval_rank_U{U}'] = dict(us=round(us, 1), users_per_s=round(U / us * 1e6, 1), table_MB=round(table_bytes / 1e6, 1),
                                      streamed_GB_per_s=round(streamed / us / 1e3, 1), flops_TF=round(2.0 * U * (a.items + 1) * E / us / 1e6, 2),
                                      note='table is 16.8 MB: Infinity-Cache resident; bound = L2/MALL bandwidth and the fp32 MFMA rate (157 TF)')
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
