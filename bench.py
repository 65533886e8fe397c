#!/usr/bin/env python
"""bench.py -- user-sequences/sec of the adapter-tuned TransRec training step on MI355X.

Workload (BASELINE.json configs[1]): SASRec + BERT-base + Houlsby adapters (width 64 in BERT, 16 in SASRec),
title length 30, 21 + 21 item slots per user (seq_len 23 raw history), bf16 storage / fp32 accumulate,
dropout ON (train mode), fused Adam on the adapter tensors, one RCCL all-reduce of the flat adapter-gradient
buffer per step when --gpus > 1.  Synthetic data per SURVEY.md section 8(d): seed 123456, 65 536 items,
canonical dense titles (30 tokens), every user a full 23-item history.  Random-init weights of the
BERT-base geometry (no checkpoints in the image).

A "step" = the PUBLIC path of the drop-in boundary, exactly what adapter4rec_amd/run.py's loop executes (reference:
Downstream/Text/run.py:595-600): optimizer.zero_grad(); loss = model(sample_items, log_mask, local_rank) through FlatDDP;
loss.backward() (native backward straight into the flat gradient buffer + ONE RCCL all-reduce of it); optimizer.step()
(FusedAdam) -- on one batch already resident in HBM.

`--gpus N` with N > 1 and no launcher environment starts N ranks itself (one process per GPU, torch.distributed.run on
127.0.0.1) BEFORE anything touches the GPU; under a launcher (WORLD_SIZE set) it is one of the ranks.  It exits non-zero if
it ends up with a rank count other than N.  The JSON line carries `rccl_ranks` (the result of an actual all-reduce of ones),
the all-reduce payload and its measured time.
Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the bf16 MFMA GEMM), measured with HIP events
in an instrumented pass run after the timed region; `cpu_baseline` times the CPU oracle (oracle/ref_cpu.py) on a
bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense fp8 (block-scaled MFMA), same guide; BASELINE.md section 4 prices configs[4] against it
# algorithmic GFLOP per user-sequence (12 S 42 [2 x 14 155 776 + 3 (4SH + f_ad)], + patch embedding for images)
GFLOP_PER_USER = {'bert_houlsby': 450.1, 'roberta_pfeiffer_cpc': 441.2, 'vit_lora': 3005.9 + 9.7, 'mae_compacter': 754.8 + 9.7,
                  # full fine-tuning: forward + dgrad + wgrad of every dense product, attention 3x: 12 S 42 [3 x 14 155 776 + 3 x 4SH]
                  'bert_pretrain': 646.3,
                  # 12 S 42 [3 x 14 155 776 + 3 x 4SH] at S = 50 + the patch projection forward and weight gradient
                  'mae_pretrain': 1081.8 + 2 * 9.7 / 4}
SEED = 123456


def make_args(batch, dtype):
    return argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=['title'], word_embedding_dim=768,
        bert_model_load='bert_base_uncased', bert_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1,
        adapter_activation='RELU', hypercomplex_division=4, phm_init_range=1e-4, adapter_type='houslby', is_serial='True',
        adding_adapter_to='all', arch='sasrec', compute_dtype=dtype, batch_size=batch,
        fine_tune_lr=5e-5, lr=1e-4, adapter_bert_lr=1.5e-4, adapter_sasrec_lr=1.5e-4)


WORKLOADS = {
    # name: (BASELINE.json config, default users/GPU, description, data)
    'bert_houlsby': ('configs[1]', 32, 'MIND-shape SASRec+BERT-base+Houlsby adapter train step (fwd+bwd+allreduce+Adam), dropout on',
                     'synthetic (seed 123456, 65536 items, 30-token titles, full 23-item histories; random-init BERT-base)'),
    'roberta_pfeiffer_cpc': ('configs[3]', 32, 'Adressa-shape CPC+RoBERTa-base+Pfeiffer adapter (relu) train step, dropout on',
                             'synthetic (seed 123456, 65536 items, 30-token titles, vocab 50265, pad id 1; random-init RoBERTa-base)'),
    'vit_lora': ('configs[2]', 8, 'HM-shape SASRec+ViT-B/16+LoRA r=8 (q, v) train step from uint8 224x224 images resident in HBM',
                 'synthetic (seed 123456, uint8 images U{0..255} [336, 224, 224, 3] per step; random-init ViT-B/16)'),
    'mae_compacter': ('configs[4]', 8, 'Amazon-shape SASRec+ViT-MAE-base (75 % masked, 50 tokens)+Compacter train step from uint8 images',
                      'synthetic (seed 123456, uint8 images, on-device masking noise; random-init ViT-MAE-base)'),
    # not a BASELINE.json config: the reference's OTHER half (Pretraining/Text/script/sm_base_sasrec.py: nothing frozen, no adapters, B = 32),
    # SURVEY 8(f) n3 -- every backbone weight gradient runs (a4r_gemm_tn), Adam over ~110 M parameters
    'mae_pretrain': ('Pretraining/CV (SURVEY 8f n3)', 8, 'HM-shape SASRec+ViT-MAE-base (75 % masked, 50 tokens) FULL fine-tuning train step (Pretraining/CV/script/sm_vit_sasrec.py: nothing frozen) from uint8 images',
                     'synthetic (seed 123456, uint8 images, on-device masking noise; random-init ViT-MAE-base)'),
    'bert_pretrain': ('Pretraining/Text (SURVEY 8f n3)', 32, 'MIND-shape SASRec+BERT-base FULL fine-tuning train step (--fine_tune_to all, no adapters), dropout on',
                      'synthetic (seed 123456, 65536 items, 30-token titles, full 23-item histories; random-init BERT-base)'),
}


def make_cv_args(batch, dtype, workload):
    a = argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        CV_model_load='vit-mae-base' if workload in ('mae_compacter', 'mae_pretrain') else 'vit-base-patch16-224', CV_resize=224,
        cv_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1, adapter_activation='RELU',
        hypercomplex_division=4, phm_init_range=1e-4, adapter_type='compacter' if workload == 'mae_compacter' else ('none' if workload == 'mae_pretrain' else 'lora'),
        is_serial='True', adding_adapter_to='None' if workload == 'mae_pretrain' else 'all', arch='sasrec', compute_dtype=dtype, batch_size=batch, lora_r=8, lora_r_sasrec=4,
        fine_tune_lr=1e-5, lr=1e-3, adapter_cv_lr=5e-4, adapter_sasrec_lr=1e-4)
    return a


def build_cv_model(args, device):
    from adapter4rec_amd.cv import Model, ViTForImageClassification, ViTMAEModel
    from adapter4rec_amd.cv.inject import inject_adapters, optimizer_groups
    from adapter4rec_amd.inject import freeze_all
    from adapter4rec_amd.optim import FusedAdam
    torch.manual_seed(SEED)
    if 'mae' in args.CV_model_load:
        net = ViTMAEModel()
    else:
        net = ViTForImageClassification(num_labels=args.embedding_dim)      # classifier swapped for Linear(768, 64), run_adapter.py:291-296
        torch.nn.init.xavier_normal_(net.classifier.weight)
    model = Model(args, 8192, True, net)
    if 'None' in args.adding_adapter_to:                      # Pretraining/CV: nothing frozen (HF's fixed sin-cos position table stays as constructed)
        pass
    else:
        freeze_all(model)
    model = inject_adapters(model, args)
    model.to(device)
    model.train()
    opt = FusedAdam(optimizer_groups(model, args))
    return model, opt


def synth_image_batches(batch, n_batches, device, seed, ragged=False, host_mask=True):
    """uint8 HWC images, 21 + 21 slots per user; the last negative slot is never filled (dataset.py:94-105).
    ragged (--ragged-histories): train lengths ~ U{2..21}, the pad slots of a short user stay ZERO images in the positive and the negative slot
    (Build_Lmdb_Dataset.__getitem__, Downstream/CV/data_utils/dataset.py:85-113), log_mask = [0] * pad + [1] * (len - 1), on the host unless host_mask
    is False (the A/B: every slot encoded)."""
    g = torch.Generator(device=device).manual_seed(seed)
    gh = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        img = torch.randint(0, 256, (batch, 21, 2, 224, 224, 3), generator=g, device=device, dtype=torch.uint8)
        img[:, -1, 1] = 0
        lm = torch.ones(batch, 20)
        if ragged:
            lens = torch.randint(2, 22, (batch,), generator=gh)
            for b in range(batch):
                pad = 21 - int(lens[b])
                img[b, :pad] = 0
                lm[b, :pad] = 0
        out.append((img.view(-1, 224, 224, 3), lm if (ragged and host_mask) else lm.to(device)))
    return out


def synth_content(n_items, g):
    """item_content [n_items + 1, 60]: [101, t_1..t_28, 102] || ones; item 0 = zeros (SURVEY.md 8(d) canonical variant)."""
    c = torch.zeros(n_items + 1, 60, dtype=torch.int64)
    c[1:, 1:29] = torch.randint(1000, 30000, (n_items, 28), generator=g)
    c[1:, 0] = 101
    c[1:, 29] = 102
    c[1:, 30:] = 1
    return c


def real_shapes():
    """tests/golden/real_shapes.json (data, written by tools/gen_golden_r6.py from the reference's readers on the files it ships): counts of attended
    tokens per Adressa title (bert_base_uncased, --num_words_title 30) and of training-history items per Amazon user (--max_seq_len 20)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'real_shapes.json')) as f:
        sh = json.load(f)
    return (torch.tensor(sh['title_tokens']['counts'], dtype=torch.float64), torch.tensor(sh['history_items']['counts'], dtype=torch.float64), sh)


def synth_batches(content, n_items, batch, n_batches, g, ragged=False, hist_counts=None):
    """Full 23-item histories: train seq = 21 items, log_mask = ones(20); one uniformly sampled negative per position.
    ragged (--ragged-histories, reported separately): train lengths ~ U{2..21}, left-padded with item 0 in the positive AND the negative slot, log_mask =
    [0] * pad + [1] * (len - 1) -- BuildTrainDataset.__getitem__, Downstream/Text/data_utils/dataset.py:24-49."""
    out = []
    for _ in range(n_batches):
        seqs = torch.stack([torch.randperm(n_items, generator=g)[:21] + 1 for _ in range(batch)])        # [B, 21]
        negs = torch.randint(1, n_items + 1, (batch, 21), generator=g)
        negs[:, -1] = 0
        if ragged:
            # --real-shaped: lengths drawn from the real histogram (index = items of the train sequence) instead of U{2..21}
            lens = torch.multinomial(hist_counts, batch, replacement=True, generator=g) if hist_counts is not None else torch.randint(2, 22, (batch,), generator=g)
            lm = torch.zeros(batch, 20)
            for b in range(batch):
                pad = 21 - int(lens[b])
                seqs[b, :pad], negs[b, :pad] = 0, 0
                lm[b, pad:] = 1
            ids = torch.stack([seqs, negs], 2).view(-1)
            out.append((content[ids].contiguous(), lm))
            continue
        ids = torch.stack([seqs, negs], 2).view(-1)                                                      # [B*21*2]
        out.append((content[ids].contiguous(), torch.ones(batch, 20)))
    return out


def build_model(args, device, roberta=False):
    from adapter4rec_amd.inject import freeze_all, inject_adapters, optimizer_groups
    from adapter4rec_amd.model import BERT_BASE, ROBERTA_BASE, BertBackbone, Model, ModelCPC
    from adapter4rec_amd.optim import FusedAdam
    torch.manual_seed(SEED)
    model = (ModelCPC if args.arch == 'cpc' else Model)(args, 65536, True, BertBackbone(ROBERTA_BASE if roberta else BERT_BASE))
    if 'None' in args.adding_adapter_to:                      # Pretraining/: nothing frozen but the pooler (run.py:317-319)
        for n, p in model.named_parameters():
            p.requires_grad = 'pooler' not in n
    else:
        freeze_all(model)
    model = inject_adapters(model, args)
    model.to(device)
    model.train()
    opt = FusedAdam(optimizer_groups(model, args))
    return model, opt


class GemmProbe:
    """HIP-event timing of every a4r_gemm_nt launch (instrumented pass only)."""

    def __init__(self, L):
        self.L, self.real, self.rec = L, L.gemm_nt, []

    def __enter__(self):
        def wrapped(A, B, Cout, *a, **k):
            M = k.get('M') or A.shape[0]
            ad = (k.get('act', 0), k.get('dact', 0))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.real(A, B, Cout, *a, **k)
            e1.record()
            N, K = B.shape[0], B.shape[1]
            # mirror of a4r_gemm_nt's dispatch (a4r_gemm.hip): one-K-tile products and N = 64 have their own streaming kernels
            bf16_in = A.dtype == torch.bfloat16
            if bf16_in and K == 64 and N >= 256 and N % 128 == 0 and M % 64 == 0 and Cout.dtype == torch.bfloat16 and not ad[1]:
                tile = ('skinnyk',)
            elif bf16_in and N == 64 and M % 64 == 0:
                tile = ('skinny64',)
            elif M % 256 == 0 and N % 256 == 0 and (K * A.element_size()) % 128 == 0:
                # + the epilogue instantiation (a4r_gemm256.hip dispatch_same: pieces a launch carries, bit 1 dropout, 2 R1, 4 R2, 8 C2;
                # -1 = the all-purpose instantiation) -- each is its own row in a rocprofv3 summary
                names = ('bias', 'C2', 'R1', 'R2', 'Pre', 'act', 'dact', 'alpha', 'drop_p')
                kw = dict(zip(names, a)); kw.update(k)
                m = (1 if kw.get('drop_p', 0.0) > 0 else 0) | (2 if kw.get('R1') is not None else 0) | (4 if kw.get('R2') is not None else 0) | (8 if kw.get('C2') is not None else 0)
                inst = {(0, 0): (0, 1, 2, 3), (self.L.ACT_GELU, 0): (8, 72), (0, self.L.DACT_MUL_Q8): (0, 32), (0, self.L.DACT_MUL): (0,)}
                if ad == (self.L.ACT_GELU, 0) and kw.get('c2_deriv') == 'q8':
                    m |= 64                                  # (the second output is the 8-bit derivative: known at compile time since round 4)
                if A.dtype == torch.bfloat16 and ad == (0, self.L.DACT_MUL_Q8) and kw.get('q8_tiled'):
                    m |= 32                                  # (tile-native derivative: the instantiation that requests it in front of the K loop)
                if A.dtype == torch.uint8:                   # e4m3 operands (+ 16: the output leaves as e4m3 too, a4r_gemm_t.c_fp8)
                    inst = {(0, 0): (0,), (self.L.ACT_GELU, 0): (8, 72, 88), (0, self.L.DACT_MUL_Q8): (16,)}
                    m |= 16 if kw.get('c_fp8') else 0
                ef = m if (A.dtype in (torch.bfloat16, torch.uint8) and Cout.dtype in (torch.bfloat16, torch.uint8) and m in inst.get(ad, ())) else -1
                tile = (256,) + ad + (ef,)
            else:
                tile = (128 if N % 128 == 0 else 64,)
            self.rec.append((str(A.dtype), 'torch.bfloat16' if Cout.dtype == torch.uint8 else str(Cout.dtype), tile, M, N, K, e0, e1))      # (an e4m3 C belongs to the bf16-storage step)
        self.L.gemm_nt = wrapped
        # the instrumented steps issue every launch from Python: the one-call-per-layer sequencer (a4r_encoder_layer_fwd / _bwd, csrc/a4r_layer.hip) enqueues the
        # SAME launches with the same arguments from C, where this wrapper cannot put events around them (tests/test_layer_call_gpu.py: bit-identical)
        from adapter4rec_amd.engine import TransRecEngine
        self._layer_call, TransRecEngine.LAYER_CALL = TransRecEngine.LAYER_CALL, False
        return self

    def __exit__(self, *exc):
        from adapter4rec_amd.engine import TransRecEngine
        self.L.gemm_nt = self.real
        TransRecEngine.LAYER_CALL = self._layer_call

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for da, dc, tile, M, N, K, e0, e1 in self.rec:
            key = (da, dc, tile)
            f, t, n = agg.get(key, (0.0, 0.0, 0))
            agg[key] = (f + 2.0 * M * N * K, t + e0.elapsed_time(e1) * 1e-3, n + 1)
        return agg

    def by_shape(self):
        out = {}
        for da, dc, tile, M, N, K, e0, e1 in self.rec:
            k = f'{tile[0]}:{M}x{N}x{K}'
            f, t, n = out.get(k, (0.0, 0.0, 0))
            out[k] = (f + 2.0 * M * N * K, t + e0.elapsed_time(e1) * 1e-3, n + 1)
        return {k: dict(launches=v[2], avg_us=round(v[1] / v[2] * 1e6, 1), tflops=round(v[0] / v[1] / 1e12, 1)) for k, v in out.items()}


RESULTS_CHANGING_KNOBS = ('A4R_DEBUG_SKIP_WGRAD', 'A4R_TN256_DBG')     # timing-only switches of tools builds: gradients wrong / missing by design


def env_knobs():
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith('A4R_')}


def lib_id():
    """path (relative to the repo) and sha256[:16] of the kernel library this process loaded"""
    import hashlib
    from adapter4rec_amd import _lib
    with open(_lib.LIB_PATH, 'rb') as f:
        h = hashlib.sha256(f.read()).hexdigest()[:16]
    return {'path': os.path.relpath(_lib.LIB_PATH, ROOT), 'sha16': h}


def host_threads():
    """Threads the CPU baseline may really use: CPU affinity, capped by the cgroup quota and by 32
    (an over-subscribed pool -- 256 threads on a quota of a few cores -- ran 20x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(device, sample_users=8):
    """BASELINE.json configs[0] (B = 8, BERT-base + Houlsby, fp32, dropout off): one training step (fwd + bwd + Adam) of the CPU
    oracle on the host cores, timed; then THE SAME batch and weights through the HIP path in its fp32 and bf16 modes and the
    max-abs differences of loss / scores / adapter gradients against the oracle.  The oracle is the checker and
    the reported baseline here, never the thing measured as `value`."""
    from oracle import ref_cpu as R
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    torch.set_num_threads(host_threads())

    def fresh(dtype):
        args = make_args(sample_users, dtype)
        torch.manual_seed(SEED)
        model = Model(args, 65536, True, BertBackbone(BERT_BASE))
        freeze_all(model)
        return inject_adapters(model, args)
    model = fresh('fp32')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(SEED)
    content = synth_content(4096, g)
    items, mask = synth_batches(content, 4096, sample_users, 1, g)[0]
    cfg = dict(R.DEFAULT_CFG)
    lrs = dict(fine_tune_lr=5e-5, lr=1e-4, adapter_bert_lr=1.5e-4, adapter_sasrec_lr=1.5e-4)
    def one_step():
        out, grads = R.loss_and_grads(sd, trainable, items, mask, cfg)
        for k in trainable:                                     # Adam, 4 lr groups (run.py:505-529)
            p = sd[k].clone()
            R.adam_step(p, grads[k], torch.zeros_like(p), torch.zeros_like(p), 1, R.lr_group(k, lrs))
        return out, grads
    one_step()                                                  # 1 warm-up + 3 timed steps (allocator, thread pool, page-in)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out, grads = one_step()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)            # `value` is the MEAN; the fastest step is reported next to it (host cores are shared: the spread is large)
    ref_loss = float(out['loss'].detach())
    diff = {}
    for dtype in ('fp32', 'bf16'):
        m = fresh(dtype).to(device)
        m.eval()                                                # dropout off, like the oracle
        loss = m(items.to(device), mask.to(device), device)
        pos, neg = m._engine().scores()
        loss.backward()
        gerr, worst = 0.0, ''
        for n, p in m.named_parameters():
            if p.requires_grad:
                r = grads[n]
                e = float((p.grad.detach().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30))
                if e > gerr:
                    gerr, worst = e, n
        diff[dtype] = dict(loss_abs=abs(float(loss) - ref_loss),
                           scores_max_abs=float(max((pos.cpu() - out['pos_score'].detach()).abs().max(), (neg.cpu() - out['neg_score'].detach()).abs().max())),
                           grad_max_err_rel_to_tensor_max=gerr, worst_grad=worst)
        del m
    torch.cuda.empty_cache()
    return dict(value=sample_users / dt, value_best_step=sample_users / min(times), step_seconds=[round(t, 2) for t in times],
                unit='user-sequences/sec', cores=torch.get_num_threads(), kind='port',
                sample=f'1 warm-up + 3 timed train steps (fwd+bwd+Adam) of oracle/ref_cpu.py, B={sample_users} users ({sample_users * 42} items x 30 tokens), '
                       f'BERT-base+Houlsby fp32, dropout off (BASELINE.json configs[0]): ' + ' / '.join(f'{t:.1f}' for t in times) + ' s',
                loss=ref_loss, diff=diff,
                diff_note='HIP path vs the CPU oracle on the identical B=8 batch and weights (random-init BERT-base, dropout off): '
                          '|loss| difference, max |pos/neg score| difference, worst max|dg|/max|g| over the adapter gradients')


def self_launch(a, argv):
    """--gpus N > 1 without a launcher: become N ranks (the reference starts one process per GPU,
    Downstream/Text/script/adapter_houlsby.py:58-59, run.py:685).  This parent never touches the GPU: it starts
    `python -m torch.distributed.run` on 127.0.0.1 as a CHILD process and exits with its code."""
    import socket
    import subprocess
    # (no device query here: the launcher holds no GPU state at all; every rank checks `world > torch.cuda.device_count()` itself in main())
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def lockstep_fields(flat_p, order_hash, world, rank, device):
    """Per-rank proof of lock-step for a SCALE record (the reference's DDP keeps replicas identical by construction,
    run.py:503,584): every rank's exact checksum of the flat parameter buffer after the last step (int64 sum of the fp32 bit patterns: replicas must
    agree BIT FOR BIT) and the fingerprint of the order in which it launched its gradient chunks (FlatDDP.last_order_hash)."""
    import torch.distributed as dist
    bits = int(flat_p.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())
    mine = torch.tensor([bits, int(order_hash)], dtype=torch.int64, device=device)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    sums, orders = [int(t[0].item()) for t in allr], [int(t[1].item()) for t in allr]
    return {'replica_param_checksums': [f'{v & 0xFFFFFFFFFFFFFFFF:016x}' for v in sums], 'replicas_bit_identical': len(set(sums)) == 1,
            'chunk_order_hashes': [f'{v:016x}' for v in orders], 'chunk_order_identical': len(set(orders)) == 1}


def control_only(a, world, rank):
    """A4R_BENCH_CONTROL_ONLY=1 (tests/test_bench_launch.py, no GPU): the launch / rendezvous / rank-accounting control flow of
    this file with the gloo backend and no compute -- rank 0 prints the JSON skeleton with value null."""
    import torch.distributed as dist
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
    ranks = int(t.item())
    if ranks != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but the all-reduce saw {ranks} rank(s)')
    # the lock-step fields of a real run (replica checksums, chunk-order hashes), here over a stand-in buffer every rank fills alike
    lock = lockstep_fields(torch.arange(1000, dtype=torch.float32) * 0.5, 12345, world, rank, torch.device('cpu'))
    if rank == 0:
        print(json.dumps({'metric': 'user-sequences/sec, seq_len=23 BERT+SASRec+Adapter', 'value': None, 'unit': 'user-sequences/sec',
                          'n_gpus': world, 'rccl_ranks': ranks, 'backend': os.environ.get('A4R_BENCH_BACKEND', 'nccl'),
                          'control_only': True, **lock}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)          # 20 warm-up + 100 timed steps
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=0, help="users per GPU per step (default: the reference's 32 for text, 8 for images)")
    ap.add_argument('--workload', default='bert_houlsby', choices=list(WORKLOADS),
                    help="bert_houlsby = the configuration BASELINE.json's metric is quoted on; the others are its remaining configs")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp8'],
                    help='fp8: bf16 storage + OCP e4m3 operands (block-scaled MFMA rate) for the frozen encoder\'s qkv / attention-output / FFN forward GEMMs and the FFN dgrads, '
                         'text and image towers (BASELINE.json quotes the headline in bf16: the default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-images', action='store_true',
                    help='image workloads: ALSO time the step with its uint8 batch coming from PINNED HOST memory every step (H2D on a copy stream, '
                         'overlapped with the previous step: the LMDB -> GPU pipeline of SURVEY 8(d)); reported next to `value`, which stays the '
                         'HBM-resident figure')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--ragged-histories', action='store_true',
                    help='NOT the canonical benchmark: user histories of 2 .. 21 items (left-padded with item 0 / zero images), log_mask handed over on the host; '
                         'the engine does not encode the pad slots')
    ap.add_argument('--ragged-device-mask', action='store_true', help='with --ragged-histories: log_mask on the device (every slot encoded: the A/B)')
    ap.add_argument('--short-titles', action='store_true',
                    help="text workloads, NOT the canonical benchmark: SURVEY 8(d)'s 'realistic' titles of n ~ U{6..20} tokens (30-token rows, the rest pad); the "
                         'batch is handed over on the HOST every step (as run.py does) and the step runs on the longest title of the batch')
    ap.add_argument('--real-shaped', action='store_true',
                    help='text workloads, NOT the canonical benchmark: title lengths drawn from the histogram of the 20 373 real Adressa titles (mean 11.6 '
                         'tokens) and history lengths from the 21 153 real Amazon users (mean 4.1 of 21 slots) that the reference ships '
                         '(tests/golden/real_shapes.json); implies --short-titles --ragged-histories (host hand-over, titles packed, pad slots not encoded)')
    ap.add_argument('--short-titles-device', action='store_true', help='with --short-titles: batches resident on the device (30 tokens per item: the A/B)')
    ap.add_argument('--residual-dtype', default='bf20', choices=['bf16', 'fp32', 'bf24', 'bf20'],
                    help="text towers: --residual_dtype of parameters.py.  Default bf20 = the product's default since round 6 (the residual stream between sub-layers "
                         "as bf16 + a nibble per element: at least the accuracy of the reference's autocast path, +1.6 %% on the headline step); bf24: a byte per "
                         "element (+2.4 %%); bf16 = round 5's stream; recorded in config.residual_dtype")
    ap.add_argument('--gemm-variant', type=int, default=-1, help='A/B knob of a4r_gemm_variant (include/a4r.h); default: the library default')
    a = ap.parse_args()
    if a.real_shaped:
        a.short_titles = a.ragged_histories = True
    # Every A4R_* variable of the environment goes into the JSON line (`env_knobs`): a record made with an A/B knob set says so itself.
    # Knobs that make results WRONG on purpose only exist in tools-only builds (make DEBUG_KNOBS=1, tools/*.sh -D builds under tools/_ab/);
    # bench.py refuses to run with one of them set, whatever library is loaded.
    refused = [k for k in env_knobs() if k in RESULTS_CHANGING_KNOBS or k.startswith('A4R_DEBUG_')]
    if refused:
        raise SystemExit(f'bench.py: refusing to run with results-changing knob(s) set: {refused}')

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a, sys.argv[1:])                           # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'bench.py --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a rank count that did not run')
    import torch.distributed as dist
    control = bool(os.environ.get('A4R_BENCH_CONTROL_ONLY'))
    backend = os.environ.get('A4R_BENCH_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; gloo only for the control-flow test
    if control:
        if world > 1:
            dist.init_process_group(backend, init_method='env://')
        return control_only(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    oversub = bool(os.environ.get('A4R_BENCH_OVERSUBSCRIBE'))     # testing only: several ranks on one GPU (gloo; RCCL refuses duplicate GPUs)
    if world > torch.cuda.device_count() and not oversub:
        raise SystemExit(f'{world} ranks but {torch.cuda.device_count()} GPU(s): one process per GPU')
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group(backend, init_method='env://', device_id=device)
        else:
            dist.init_process_group(backend, init_method='env://')
    rccl_ranks = 1
    if world > 1:                                              # an actual collective over the group: the rank count RCCL sees
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        rccl_ranks = int(t.item())
        if rccl_ranks != a.gpus:
            raise SystemExit(f'--gpus {a.gpus} but the all-reduce saw {rccl_ranks} rank(s)')

    from adapter4rec_amd import _lib as L
    if a.gemm_variant >= 0:
        L.gemm_variant(a.gemm_variant)
    wl = a.workload
    a.batch = a.batch or WORKLOADS[wl][1]
    image = wl in ('vit_lora', 'mae_compacter', 'mae_pretrain')
    if image:
        args = make_cv_args(a.batch, a.dtype, wl)
        model, opt = build_cv_model(args, device)
        batches = synth_image_batches(a.batch, 4 if a.ragged_histories else 2, device, SEED + rank, ragged=a.ragged_histories, host_mask=not a.ragged_device_mask)
    else:
        args = make_args(a.batch, a.dtype)
        args.residual_dtype = a.residual_dtype
        if wl == 'roberta_pfeiffer_cpc':
            args.adapter_type, args.adapter_activation, args.arch, args.bert_model_load = 'pfeiffer', 'relu', 'cpc', 'roberta_base'
        if wl == 'bert_pretrain':
            args.adapter_type, args.adding_adapter_to = 'none', 'None'
        model, opt = build_model(args, device, roberta=(wl == 'roberta_pfeiffer_cpc'))
        g = torch.Generator().manual_seed(SEED + rank)            # users are sharded: every rank draws its own users
        gc = torch.Generator().manual_seed(SEED)
        content = synth_content(65536, gc)
        if wl == 'roberta_pfeiffer_cpc':                           # <s> ... </s>, pad id 1
            content[1:, 0], content[1:, 29] = 0, 2
            content[1:, 1:29] = torch.randint(3, 50265, (65536, 28), generator=gc)
        # --ragged-histories: log_mask stays on the HOST, as run.py hands it over (the engine reads the pad slots from it; --ragged-device-mask: on the device,
        # i.e. every slot encoded -- the A/B of that path)
        if a.short_titles:                                         # titles of 6 .. 20 tokens: [CLS] t .. [SEP] pad ...
            lens = torch.randint(6, 21, (content.shape[0],), generator=gc)
            if a.real_shaped:
                lens = torch.multinomial(real_shapes()[0], content.shape[0], replacement=True, generator=gc).clamp_(min=2)
            col = torch.arange(30)[None, :]
            sep = 2 if wl == 'roberta_pfeiffer_cpc' else 102
            padid = 1 if wl == 'roberta_pfeiffer_cpc' else 0
            ids, am = content[:, :30].clone(), content[:, 30:].clone()
            ids = torch.where(col == (lens[:, None] - 1), torch.full_like(ids, sep), ids)
            ids = torch.where(col >= lens[:, None], torch.full_like(ids, padid), ids)
            am = (col < lens[:, None]).long()
            ids[0], am[0] = content[0, :30], content[0, 30:]
            content = torch.cat([ids, am], 1)
        batches = [((i.pin_memory() if (a.short_titles and not a.short_titles_device) else i.to(device)),
                    m.pin_memory() if ((a.ragged_histories and not a.ragged_device_mask) or (a.short_titles and not a.short_titles_device)) else m.to(device))      # (host batches are PINNED, as run.py's DataLoader(pin_memory=True) hands them over)
                   for i, m in synth_batches(content, 65536, a.batch, 4, g, ragged=a.ragged_histories, hist_counts=real_shapes()[1] if a.real_shaped else None)]
    from adapter4rec_amd.ddp import FlatDDP
    ddp = FlatDDP(model, device_ids=[local], output_device=local)     # broadcasts rank 0's state once (run.py:503); frozen weights never move again
    inner = getattr(model, 'model', model)
    eng = inner._engine()                                      # (FlatDDP's broadcast invalidates the packed copies: this is the engine that runs)

    def step(i, api=False):
        items, mask = batches[i % len(batches)]
        if api:                                                # the engine API underneath (instrumented pass on rank 0: no collective)
            eng.flat_g.zero_()
            loss = eng.train_forward(items.to(device), mask.to(device))
            eng.train_backward(into_flat_grad=True)
            opt.step()
            return loss
        opt.zero_grad()
        loss = ddp(items, mask, local)                         # run.py:597-600
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # host time to ENQUEUE a step (nothing in a step waits for the device): a short burst behind the timed region, from an idle device -- over the K timed
    # steps the host is throttled by the full launch queue and its clock only mirrors the device's.  host >= device means the step is host-bound.
    nb = min(8, a.steps)
    t1 = time.perf_counter()
    for i in range(nb):
        step(a.warmup + a.steps + i)
    dt_host = (time.perf_counter() - t1) / nb * a.steps
    torch.cuda.synchronize()
    rank_ms = [round(dt / a.steps * 1e3, 3)]
    if world > 1:
        # every rank's own clock over the same K steps (they end at a barrier, so the spread is what each rank measured between ITS
        # synchronisation points): the first SCALE record shows by itself whether one rank / link lags; `value` uses the MAX
        each = torch.zeros(world, device=device, dtype=torch.float64)      # (an all-reduce of one-hot rows: every backend has it for device tensors)
        each[rank] = dt
        dist.all_reduce(each)
        rank_ms = [round(float(x) / a.steps * 1e3, 3) for x in each.tolist()]
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.detach())
    assert loss_val == loss_val, 'NaN loss'
    lock = lockstep_fields(eng.flat_p, getattr(ddp, 'last_order_hash', 0), world, rank, device)     # (a collective: every rank, before rank 0 goes on alone)
    if world > 1 and not (lock['replicas_bit_identical'] and lock['chunk_order_identical']):
        raise SystemExit(f'bench.py: the replicas left lock-step: {lock}')

    ar_us = None
    if world > 1:                                              # the exchange step on its own: 20 all-reduces of the flat gradient buffer
        scratch = torch.zeros_like(eng.flat_g)
        for _ in range(3):
            ddp.average_(scratch)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ddp.average_(scratch)
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / 20 * 1e3], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ar_us = round(float(t.item()), 1)

    host_leg = None
    if image and a.host_images:
        # The input pipeline of Downstream/CV/data_utils/dataset.py:85-113 ends in a host tensor per batch; here that tensor is the raw uint8
        # HWC batch (resize / normalise / patchify run on the GPU).  Two pinned host batches, two device buffers, one copy stream: the copy of
        # batch i + 1 is enqueued as soon as step i has been enqueued and waits (event) for step i - 1, the last reader of its buffer.
        K = a.steps
        pinned = [b[0].cpu().pin_memory() for b in batches]
        dev_buf = [torch.empty_like(batches[0][0]) for _ in range(2)]
        copy_s = torch.cuda.Stream(device=device)
        ready = [torch.cuda.Event() for _ in range(2)]           # H2D of buffer j finished
        freed = [torch.cuda.Event() for _ in range(2)]           # the step that read buffer j has been enqueued in full
        main_s = torch.cuda.current_stream(device)
        mask_dev = batches[0][1]

        def put(i):
            j = i % 2
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(freed[j])
                dev_buf[j].copy_(pinned[i % len(pinned)], non_blocking=True)
                ready[j].record(copy_s)

        def host_step(i):
            j = i % 2
            main_s.wait_event(ready[j])
            opt.zero_grad()
            loss = ddp(dev_buf[j], mask_dev, local)
            loss.backward()
            opt.step()
            freed[j].record(main_s)
            put(i + 2)
            return loss
        for j in range(2):
            freed[j].record(main_s)
        put(0)
        put(1)
        for i in range(max(3, a.warmup // 2)):
            host_step(i)
        i0 = max(3, a.warmup // 2)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0h = time.perf_counter()
        for i in range(K):
            host_step(i0 + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dth = time.perf_counter() - t0h
        if world > 1:
            t = torch.tensor([dth], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dth = float(t.item())
        # the copy on its own (no compute beside it): what the PCIe leg costs when nothing hides it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(10):
            dev_buf[0].copy_(pinned[0], non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        h2d_ms = e0.elapsed_time(e1) / 10
        nbytes = pinned[0].numel()
        host_leg = {'value_with_h2d': round(world * a.batch * K / dth, 2), 'ms_per_step_with_h2d': round(dth / K * 1e3, 3),
                    'h2d_bytes_per_step': int(nbytes), 'h2d_ms_alone': round(h2d_ms, 3), 'h2d_GBps_alone': round(nbytes / h2d_ms / 1e6, 1),
                    'overlap': 'copy stream, double-buffered pinned host -> device, batch i + 1 copied during step i',
                    'with_over_without': round((world * a.batch * K / dth) / (world * a.batch * a.steps / dt), 4)}

    roof = None
    if rank == 0 and not a.no_roofline:
        import adapter4rec_amd.engine as E
        side, E.WGRAD_STREAM = E.WGRAD_STREAM, False             # single stream here: an event pair would also time the wait for CUs that
        with GemmProbe(E.L) as probe:                            # side-stream weight-gradient kernels still hold when a GEMM is launched
            for i in range(2):                                   # rank 0 only: NO collective in here (the other ranks have moved on)
                step(a.warmup + a.steps + i, api=True)
            agg = probe.summary()
            shapes = probe.by_shape()
        E.WGRAD_STREAM = side
        tname = 'torch.float32' if a.dtype == 'fp32' else 'torch.bfloat16'
        big = lambda k: (k[0] in (tname, 'torch.uint8')) and k[1] == tname         # the step's large GEMMs (compute-dtype in and out)
        # the DOMINANT KERNEL is the tile family with most GPU time (the 256 x 256 persistent kernel, a4r_gemm256.hip); its epilogue
        # instantiations are separate rows of a rocprofv3 summary, so they are listed one by one below -- but the headline
        # `achieved` / `frac` is the WHOLE family (not the best instantiation)
        fam_t = {}
        for k, v in agg.items():
            if big(k):
                fam_t[k[2][0]] = fam_t.get(k[2][0], 0.0) + v[1]
        fam = max(fam_t, key=fam_t.get)
        rows = {k: v for k, v in agg.items() if big(k) and k[2][0] == fam}
        key = max(rows, key=lambda k: rows[k][1])                                   # the instantiation with most GPU time
        fp8_f = sum(v[0] for k, v in agg.items() if k[0] == 'torch.uint8')
        fp8_t = sum(v[1] for k, v in agg.items() if k[0] == 'torch.uint8')
        f = sum(v[0] for v in rows.values())
        t = sum(v[1] for v in rows.values())
        n = sum(v[2] for v in rows.values())
        ach = f / t / 1e12
        total_f = sum(v[0] for v in agg.values())
        total_t = sum(v[1] for v in agg.values())
        # fp32: the exact-fp32 MFMA (157 TF); fp8: BASELINE.md section 4 prices configs[4] against the 5 PF dense fp8 peak (only the
        # block-scaled MFMA reaches it; the launches that still run in bf16 are priced against it too -- that is the config's ceiling)
        peak = {'fp32': 157.3, 'bf16': MFMA_BF16_PEAK_TFLOPS, 'fp8': MFMA_FP8_PEAK_TFLOPS}[a.dtype]

        def inst_name(k):
            if k[2][0] == 256:
                tin = 'e4m3' if k[0] == 'torch.uint8' else a.dtype
                return f'gemm_nt_256_kernel<{tin},{a.dtype if a.dtype != "fp8" else "bf16"},act={k[2][1]},dact={k[2][2]},epilogue_pieces={k[2][3]}>'
            return f'{k[2][0]}_kernel<{a.dtype},{a.dtype}>' if isinstance(k[2][0], str) else f'gemm_nt_kernel<{a.dtype},{a.dtype},{k[2][0]}>'
        table = [dict(kernel=inst_name(k), launches_per_step=v[2] // 2, avg_launch_us=round(v[1] / v[2] * 1e6, 2), tflops=round(v[0] / v[1] / 1e12, 1),
                      frac=round(v[0] / v[1] / 1e12 / peak, 4), ms_per_step=round(v[1] / 2 * 1e3, 3))
                 for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])]
        roof = dict(bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    kernel=(f'gemm_nt_256_kernel<...> (all {len(rows)} epilogue instantiations the step launches, see `instantiations`)' if fam == 256 else inst_name(key)),
                    launches_per_step=n // 2, avg_launch_us=round(t / n * 1e6, 2), flop_per_launch=f / n,
                    instantiations=table, dominant_instantiation=inst_name(key),
                    dominant_instantiation_tflops=round(rows[key][0] / rows[key][1] / 1e12, 2),
                    all_gemm_tflops=round(total_f / total_t / 1e12, 2), gemm_time_share_of_step=round(total_t / 2 / (dt / a.steps), 3),
                    # EXECUTED flops: 2MNK of every a4r_gemm_nt launch of one step (the probe above), i.e. without the work the path skips
                    # (the last layer's attention-output / FFN half runs on the CLS rows only, layer 0 has no d qkv) -- SURVEY 8(d)
                    executed_gemm_tflop_per_step=round(total_f / 2 / 1e12, 3),
                    step_tflops_per_gpu=round(total_f / 2 / (dt / a.steps) / 1e12, 1),
                    step_frac_of_peak=round(total_f / 2 / (dt / a.steps) / 1e12 / peak, 4),
                    # SURVEY's formula (42 items x 12 FULL layers, forward + 2 x backward): what a path that skips nothing would execute
                    algorithmic_step_tflops_per_gpu=round(a.batch * a.steps / dt * GFLOP_PER_USER[wl] / 1e3, 1),
                    algorithmic_step_frac_of_peak=round(a.batch * a.steps / dt * GFLOP_PER_USER[wl] / 1e3 / peak, 4))
        if fp8_t > 0:
            roof['fp8_gemm_tflops'] = round(fp8_f / fp8_t / 1e12, 2)
            roof['fp8_share_of_gemm_flops'] = round(fp8_f / total_f, 3)
        # fabric/HBM bytes per launch of that kernel: not measurable from inside the process -- taken from the committed
        # rocprofv3 PMC passes over this same command (profiles/r02_l_pmc_hbm_traffic.json says how); only when the run IS that
        # command's configuration (bert_houlsby, B=32, bf16), null otherwise.
        import glob
        pmcs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pmc_hbm_traffic.json')))
        if fam == 256 and a.dtype == 'bf16' and a.batch == 32 and wl == 'bert_houlsby' and pmcs:
            pmc = pmcs[-1]                                          # the newest committed pass over this same command
            per = {}
            for k, v in rows.items():
                mangled = f'gemm_nt_256_kernelIDF16bDF16bLi{k[2][1]}ELi{k[2][2]}ELi{k[2][3]}E'.replace('Li-1E', 'Lin1E')
                for kname, rec in json.load(open(pmc))['kernels'].items():
                    if mangled in kname:
                        per[inst_name(k)] = (rec['traffic_bytes_per_launch'], v[2])
            if len(per) == len(rows):                                # launch-weighted mean over the family, like `achieved`
                roof['traffic'] = round(sum(b * c for b, c in per.values()) / sum(c for _, c in per.values()))
                roof['traffic_per_instantiation'] = {k: b for k, (b, _) in per.items()}
                roof['traffic_source'] = (f'NOT measured in this run: profiles/{os.path.basename(pmc)}, separate rocprofv3 --pmc FETCH_SIZE / '
                                          'WRITE_SIZE passes over this same command (2 x FETCH correction of the gfx950 guide)')
        if os.environ.get('A4R_BENCH_SHAPES'):
            print(json.dumps(shapes, indent=1), file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and wl == 'bert_houlsby':
        cpu = cpu_baseline(device)

    if rank == 0:
        users = world * a.batch * a.steps
        out = {
            'metric': {'bert_houlsby': 'user-sequences/sec, seq_len=23 BERT+SASRec+Adapter', 'roberta_pfeiffer_cpc': 'user-sequences/sec, seq_len=23 RoBERTa+CPC+Pfeiffer',
                       'vit_lora': 'user-sequences/sec, seq_len=23 ViT+SASRec+LoRA', 'mae_compacter': 'user-sequences/sec, seq_len=23 MAE+SASRec+Compacter',
                       'bert_pretrain': 'user-sequences/sec, seq_len=23 BERT+SASRec full fine-tuning (Pretraining/Text)',
                       'mae_pretrain': 'user-sequences/sec, seq_len=23 MAE+SASRec full fine-tuning (Pretraining/CV)'}[wl],
            'value': round(users / dt, 2),
            'unit': 'user-sequences/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'host_enqueue_ms_per_step': round(dt_host / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': WORKLOADS[wl][3] + (' -- RAGGED histories of 2..21 items (not the canonical benchmark)' if getattr(a, 'ragged_histories', False) else '') + (' -- SHORT titles of 6..20 tokens (not the canonical benchmark)' if getattr(a, 'short_titles', False) else '')
                    + (' -- REAL-SHAPED: title lengths ~ the 20 373 Adressa titles, history lengths ~ the 21 153 Amazon users the reference ships (tests/golden/real_shapes.json)' if getattr(a, 'real_shaped', False) else ''),
            'config': {'workload': WORKLOADS[wl][2], 'baseline_config': WORKLOADS[wl][0] + (' in bf16 (run with --dtype fp8 for its fp8 encoder)' if wl == 'mae_compacter' and a.dtype != 'fp8' else ''),
                       'users_per_gpu': a.batch, 'global_batch': world * a.batch, 'seq_len': 23,
                       'tokens_per_item': eng.S, 'items_per_user': 42,
                       # item slots the model reads: Model.forward drops every user's last negative, ModelCPC.forward reads one negative only; the
                       # engine does not encode unread slots where that removes work (engine.py: _kept_rows; A4R_SKIP_UNUSED_ITEMS=0: all 42)
                       'items_encoded_per_user': (eng._kept_rows(a.batch) or 42 * a.batch) // a.batch, 'parallelism': f'dp{world}',
                       'path': 'public: optimizer.zero_grad(); FlatDDP(model)(items, mask); loss.backward(); FusedAdam.step()',
                       **({'residual_dtype': a.residual_dtype,
                           'residual_dtype_cost': 'bf20 (default): +1.6 % step time against --residual-dtype bf16 same-box, bf24 +2.4 % (profiles/r06_h_residual_bf20.txt); '
                                                  'either buys rms error <= 1.0 x the reference-under-autocast (bf16 stream: 1.2 - 1.3 x)'} if not image else {})},
            'rccl_ranks': rccl_ranks, 'allreduce_bytes_per_step': int(eng.flat_g.numel() * 4) if world > 1 else 0, 'allreduce_us': ar_us,
            'ms_per_step_ranks': rank_ms, 'ms_per_step_spread': round(max(rank_ms) - min(rank_ms), 3),
            'allreduce_overlapped': bool(world > 1 and eng.OVERLAP_ALLREDUCE and eng._grad_chunks() is not None),
            'loss': round(loss_val, 5), 'roofline': roof, 'cpu_baseline': cpu,
            'env_knobs': env_knobs(), 'lib': lib_id(), **lock,
        }
        if host_leg is not None:
            out['host_images'] = host_leg
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                 # rank 0 ran its instrumented pass alone: leave the group together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
