"""Train-step time of the headline model (BERT-base + Houlsby + SASRec, bf16, dropout on, B = 32 users) at title lengths and history lengths other
than the canonical 30 tokens / 20 positions -- the paths --num_words_title > 32 and --max_seq_len > 32 take (long attention kernels with a key mask,
causal in the user tower; Downstream/Text/parameters.py:29,44).  Not a BASELINE.json configuration: reported beside the headline, never as it.

    python tools/long_inputs_bench.py [--steps 30] [--warmup 8]

One JSON line per case: ms per step, user-sequences/s, title-token rows per second (items x tokens: the unit the encoder's cost scales with)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench as B  # noqa: E402


def content_of(n_items, W, g):
    c = torch.zeros(n_items + 1, 2 * W, dtype=torch.int64)
    c[1:, 1:W - 1] = torch.randint(1000, 30000, (n_items, W - 2), generator=g)
    c[1:, 0], c[1:, W - 1], c[1:, W:] = 101, 102, 1
    return c


def batches_of(content, n_items, batch, Lq, g, n=3):
    out = []
    for _ in range(n):
        seqs = torch.stack([torch.randperm(n_items, generator=g)[:Lq] + 1 for _ in range(batch)])
        negs = torch.randint(1, n_items + 1, (batch, Lq), generator=g)
        negs[:, -1] = 0
        ids = torch.stack([seqs, negs], 2).view(-1)
        out.append((content[ids].contiguous(), torch.ones(batch, Lq - 1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--cases', default='30x20,32x20,33x20,50x20,100x20,30x32,30x33,30x50,30x100,50x50')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for case in a.cases.split(','):
        W, T = (int(x) for x in case.split('x'))
        args = B.make_args(a.batch, 'bf16')
        args.num_words_title, args.max_seq_len = W, T
        model, opt = B.build_model(args, dev)
        g = torch.Generator().manual_seed(B.SEED)
        content = content_of(65536, W, g)
        batches = [(i.to(dev), m.to(dev)) for i, m in batches_of(content, 65536, a.batch, T + 1, g)]

        def step(i):
            items, mask = batches[i % len(batches)]
            opt.zero_grad()
            loss = model(items, mask, 0)
            loss.backward()
            opt.step()
            return loss
        for i in range(a.warmup):
            loss = step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            loss = step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        lv = float(loss.detach())
        assert lv == lv, 'NaN loss'
        n_items = a.batch * (2 * (T + 1) - 1)                       # the last negative slot is never encoded
        print(json.dumps({'title_tokens': W, 'max_seq_len': T, 'users': a.batch, 'ms_per_step': round(ms, 3), 'user_seq_per_s': round(a.batch / ms * 1e3, 1),
                          'items_per_step': n_items, 'token_rows_per_s_M': round(n_items * W / ms / 1e3, 2), 'loss': round(lv, 4)}), flush=True)
        del model, opt, batches
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
