#!/usr/bin/env python
"""bench.py -- user-sequences/sec of the adapter-tuned TransRec training step on MI355X.

Workload (BASELINE.json configs[1]): SASRec + BERT-base + Houlsby adapters (width 64 in BERT, 16 in SASRec),
title length 30, 21 + 21 item slots per user (seq_len 23 raw history), bf16 storage / fp32 accumulate,
dropout ON (train mode), fused Adam on the adapter tensors, one RCCL all-reduce of the flat adapter-gradient
buffer per step when --gpus > 1.  Synthetic data per SURVEY.md section 8(d): seed 123456, 65 536 items,
canonical dense titles (30 tokens), every user a full 23-item history.  Random-init weights of the
BERT-base geometry (no checkpoints in the image).

A "step" = forward + backward + gradient all-reduce + Adam on one batch already resident in HBM.
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
    'mae_compacter': ('configs[4], bf16 instead of fp8', 8, 'Amazon-shape SASRec+ViT-MAE-base (75 % masked, 50 tokens)+Compacter train step from uint8 images',
                      'synthetic (seed 123456, uint8 images, on-device masking noise; random-init ViT-MAE-base)'),
}


def make_cv_args(batch, dtype, workload):
    a = argparse.Namespace(
        max_seq_len=20, l2_weight=0, embedding_dim=64, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
        CV_model_load='vit-mae-base' if workload == 'mae_compacter' else 'vit-base-patch16-224', CV_resize=224,
        cv_adapter_down_size=64, adapter_down_size=16, adapter_dropout_rate=0.1, adapter_activation='RELU',
        hypercomplex_division=4, phm_init_range=1e-4, adapter_type='compacter' if workload == 'mae_compacter' else 'lora',
        is_serial='True', adding_adapter_to='all', arch='sasrec', compute_dtype=dtype, batch_size=batch, lora_r=8, lora_r_sasrec=4,
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
    freeze_all(model)
    model = inject_adapters(model, args)
    model.to(device)
    model.train()
    opt = FusedAdam(optimizer_groups(model, args))
    return model, opt


def synth_image_batches(batch, n_batches, device, seed):
    """uint8 HWC images, 21 + 21 slots per user; the last negative slot is never filled (dataset.py:94-105)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    for _ in range(n_batches):
        img = torch.randint(0, 256, (batch, 21, 2, 224, 224, 3), generator=g, device=device, dtype=torch.uint8)
        img[:, -1, 1] = 0
        out.append((img.view(-1, 224, 224, 3), torch.ones(batch, 20, device=device)))
    return out


def synth_content(n_items, g):
    """item_content [n_items + 1, 60]: [101, t_1..t_28, 102] || ones; item 0 = zeros (SURVEY.md 8(d) canonical variant)."""
    c = torch.zeros(n_items + 1, 60, dtype=torch.int64)
    c[1:, 1:29] = torch.randint(1000, 30000, (n_items, 28), generator=g)
    c[1:, 0] = 101
    c[1:, 29] = 102
    c[1:, 30:] = 1
    return c


def synth_batches(content, n_items, batch, n_batches, g):
    """Full 23-item histories: train seq = 21 items, log_mask = ones(20); one uniformly sampled negative per position."""
    out = []
    for _ in range(n_batches):
        seqs = torch.stack([torch.randperm(n_items, generator=g)[:21] + 1 for _ in range(batch)])        # [B, 21]
        negs = torch.randint(1, n_items + 1, (batch, 21), generator=g)
        negs[:, -1] = 0
        ids = torch.stack([seqs, negs], 2).view(-1)                                                      # [B*21*2]
        out.append((content[ids].contiguous(), torch.ones(batch, 20)))
    return out


def build_model(args, device, roberta=False):
    from adapter4rec_amd.inject import freeze_all, inject_adapters, optimizer_groups
    from adapter4rec_amd.model import BERT_BASE, ROBERTA_BASE, BertBackbone, Model, ModelCPC
    from adapter4rec_amd.optim import FusedAdam
    torch.manual_seed(SEED)
    model = (ModelCPC if args.arch == 'cpc' else Model)(args, 65536, True, BertBackbone(ROBERTA_BASE if roberta else BERT_BASE))
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
                tile = (256,) + ad
            else:
                tile = (128 if N % 128 == 0 else 64,)
            self.rec.append((str(A.dtype), str(Cout.dtype), tile, M, N, K, e0, e1))
        self.L.gemm_nt = wrapped
        return self

    def __exit__(self, *exc):
        self.L.gemm_nt = self.real

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


def cpu_baseline(sample_users=16):
    """The CPU oracle on a bounded sample: one training step (fwd + bwd + Adam), BERT-base + Houlsby, fp32."""
    from oracle import ref_cpu as R
    from adapter4rec_amd.inject import freeze_all, inject_adapters
    from adapter4rec_amd.model import BERT_BASE, BertBackbone, Model
    torch.set_num_threads(host_threads())
    args = make_args(sample_users, 'fp32')
    torch.manual_seed(SEED)
    model = Model(args, 65536, True, BertBackbone(BERT_BASE))
    freeze_all(model)
    model = inject_adapters(model, args)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(SEED)
    content = synth_content(4096, g)
    items, mask = synth_batches(content, 4096, sample_users, 1, g)[0]
    cfg = dict(R.DEFAULT_CFG)
    lrs = dict(fine_tune_lr=5e-5, lr=1e-4, adapter_bert_lr=1.5e-4, adapter_sasrec_lr=1.5e-4)
    t0 = time.perf_counter()
    R.train_steps(sd, trainable, [(items, mask)], cfg, lrs, 1)
    dt = time.perf_counter() - t0
    return dict(value=sample_users / dt, unit='user-sequences/sec', cores=torch.get_num_threads(), kind='port',
                sample=f'1 train step (fwd+bwd+Adam) of oracle/ref_cpu.py, B={sample_users} users ({sample_users * 42} items x 30 tokens), '
                       f'BERT-base+Houlsby fp32, dropout off, {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help="users per GPU per step (default: the reference's 32 for text, 8 for images)")
    ap.add_argument('--workload', default='bert_houlsby', choices=list(WORKLOADS),
                    help="bert_houlsby = the configuration BASELINE.json's metric is quoted on; the others are its remaining configs")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--gemm-variant', type=int, default=-1, help='A/B knob of a4r_gemm_variant (include/a4r.h); default: the library default')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    local = local % torch.cuda.device_count()                  # (one rank per GPU in real runs; lets the N > 1 control flow be exercised on one GPU)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(os.environ.get('A4R_BENCH_BACKEND', 'nccl'), init_method='env://')     # 'nccl' is RCCL on ROCm; gloo only for the control-flow test
    assert world == a.gpus or world == 1, f'--gpus {a.gpus} but WORLD_SIZE={world}'

    from adapter4rec_amd import _lib as L
    if a.gemm_variant >= 0:
        L.gemm_variant(a.gemm_variant)
    wl = a.workload
    a.batch = a.batch or WORKLOADS[wl][1]
    image = wl in ('vit_lora', 'mae_compacter')
    if image:
        args = make_cv_args(a.batch, a.dtype, wl)
        model, opt = build_cv_model(args, device)
        inner = getattr(model, 'model', model)
        eng = inner._engine()
        batches = synth_image_batches(a.batch, 2, device, SEED + rank)
    else:
        args = make_args(a.batch, a.dtype)
        if wl == 'roberta_pfeiffer_cpc':
            args.adapter_type, args.adapter_activation, args.arch, args.bert_model_load = 'pfeiffer', 'relu', 'cpc', 'roberta_base'
        model, opt = build_model(args, device, roberta=(wl == 'roberta_pfeiffer_cpc'))
        eng = model._engine()
        g = torch.Generator().manual_seed(SEED + rank)            # users are sharded: every rank draws its own users
        gc = torch.Generator().manual_seed(SEED)
        content = synth_content(65536, gc)
        if wl == 'roberta_pfeiffer_cpc':                           # <s> ... </s>, pad id 1
            content[1:, 0], content[1:, 29] = 0, 2
            content[1:, 1:29] = torch.randint(3, 50265, (65536, 28), generator=gc)
        batches = [(i.to(device), m.to(device)) for i, m in synth_batches(content, 65536, a.batch, 4, g)]
    if world > 1:                                              # DDP constructor semantics: rank 0's trainables everywhere
        dist.broadcast(eng.flat_p, 0)

    def step(i, exchange=True):
        items, mask = batches[i % len(batches)]
        eng.flat_g.zero_()
        loss = eng.train_forward(items, mask)
        eng.train_backward(into_flat_grad=True)
        if world > 1 and exchange:
            dist.all_reduce(eng.flat_g)
        opt.step(grad_scale=1.0 / world)
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
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)
    assert loss_val == loss_val, 'NaN loss'

    roof = None
    if rank == 0 and not a.no_roofline:
        import adapter4rec_amd.engine as E
        side, E.WGRAD_STREAM = E.WGRAD_STREAM, False             # single stream here: an event pair would also time the wait for CUs that
        with GemmProbe(E.L) as probe:                            # side-stream weight-gradient kernels still hold when a GEMM is launched
            for i in range(2):                                   # rank 0 only: NO collective in here (the other ranks have moved on)
                step(a.warmup + a.steps + i, exchange=False)
            agg = probe.summary()
            shapes = probe.by_shape()
        E.WGRAD_STREAM = side
        tname = 'torch.bfloat16' if a.dtype == 'bf16' else 'torch.float32'
        key = max((k for k in agg if k[0] == tname and k[1] == tname), key=lambda k: agg[k][1])     # most GPU time
        f, t, n = agg[key]
        ach = f / t / 1e12
        total_f = sum(v[0] for v in agg.values())
        total_t = sum(v[1] for v in agg.values())
        peak = MFMA_BF16_PEAK_TFLOPS if a.dtype == 'bf16' else 157.3
        roof = dict(bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    kernel=(f'gemm_nt_256_kernel<{a.dtype},{a.dtype},act={key[2][1]},dact={key[2][2]}>' if key[2][0] == 256 else (f'{key[2][0]}_kernel<{a.dtype},{a.dtype}>' if isinstance(key[2][0], str) else f'gemm_nt_kernel<{a.dtype},{a.dtype},{key[2][0]}>')), launches_per_step=n // 2,
                    avg_launch_us=round(t / n * 1e6, 2), flop_per_launch=f / n,
                    all_gemm_tflops=round(total_f / total_t / 1e12, 2), gemm_time_share_of_step=round(total_t / 2 / (dt / a.steps), 3))
        # fabric/HBM bytes per launch of that kernel: not measurable from inside the process -- taken from the committed
        # rocprofv3 PMC passes over this same command (profiles/r01_g_pmc_hbm_traffic.json says how), B=32 bf16 only.
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_g_pmc_hbm_traffic.json')
        if key[2][0] == 256 and a.dtype == 'bf16' and a.batch == 32 and os.path.exists(pmc):
            mangled = f'gemm_nt_256_kernelIDF16bDF16bLi{key[2][1]}ELi{key[2][2]}E'
            for kname, rec in json.load(open(pmc))['kernels'].items():
                if mangled in kname:
                    roof['traffic'] = rec['traffic_bytes_per_launch']
                    roof['traffic_source'] = 'profiles/r01_g_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2 x FETCH correction)'
        if os.environ.get('A4R_BENCH_SHAPES'):
            print(json.dumps(shapes, indent=1), file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and wl == 'bert_houlsby':
        cpu = cpu_baseline()

    if rank == 0:
        users = world * a.batch * a.steps
        out = {
            'metric': {'bert_houlsby': 'user-sequences/sec, seq_len=23 BERT+SASRec+Adapter', 'roberta_pfeiffer_cpc': 'user-sequences/sec, seq_len=23 RoBERTa+CPC+Pfeiffer',
                       'vit_lora': 'user-sequences/sec, seq_len=23 ViT+SASRec+LoRA', 'mae_compacter': 'user-sequences/sec, seq_len=23 MAE+SASRec+Compacter'}[wl],
            'value': round(users / dt, 2),
            'unit': 'user-sequences/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': WORKLOADS[wl][3],
            'config': {'workload': WORKLOADS[wl][2], 'baseline_config': WORKLOADS[wl][0],
                       'users_per_gpu': a.batch, 'global_batch': world * a.batch, 'seq_len': 23,
                       'tokens_per_item': eng.S, 'items_per_user': 42, 'parallelism': f'dp{world}'},
            'loss': round(loss_val, 5), 'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                 # rank 0 ran its instrumented pass alone: leave the group together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
