"""The product's counter-based dropout stream, restated for the checker.  TEST INFRASTRUCTURE ONLY (see oracle/ref_cpu.py's header).

The HIP kernels draw no random numbers: an element keeps its value iff a 16-bit lot of a hash of (seed, site, element index) reaches the
threshold round(p * 65536) (adapter4rec_amd/csrc/a4r_common.h:278-307: a4r_mix32, a4r_hash64, dropout_keep, a4r_thr16, a4r_keep_scale), and the
backward pass regenerates the mask from the same triple.  To check a TRAINING-mode step (the mode bench.py times) against the oracle, the oracle
has to multiply by the very same masks at the reference's dropout sites (torch.nn.Dropout in HF BertEmbeddings / BertSelfAttention /
BertSelfOutput / BertOutput and Downstream/Text/model/modules.py:27,40,70,104).  This file restates the hash bit for bit in numpy and the
element index every kernel family hands it:

    kind 'rows'       e = row * H + col of the logical [rows, H] tensor           GEMM epilogue (a4r_gemm_epi.h:205), LayerNorm / embedding rows
                                                                                   (a4r_rows.hip:59), fused adapter (a4r_adapter_fused.hip:538)
    kind 'rows_cls'   the LAST encoder layer's two dense dropouts: the engine runs them on the CLS rows only (engine.py: _block_forward, cls_rows),
                      so token 0 of item i is row i of an [items, H] tensor there and the other tokens (never read) keep their values
    kind 'attn_item'  item tower probabilities [items, heads, S, S], pair = item * heads + head:
                         head dim <= 16:  ((pair * S + q) * S + k)                 a4r_attn_small.hip:52
                         S <= 32:         ((pair * 32 + q) * 32 + k)               a4r_attn.hip:151
                         else:            ((pair * 256 + q) << 8) + k              a4r_attn_long.hip:151
    kind 'attn_user'  SASRec block probabilities [users, heads, T, T]:  (((user * heads + head) * 32 + q) << 5) + k       a4r_sasrec.hip:202
                      (T > 32, the long attention kernels:               (((user * heads + head) * 256 + q) << 8) + k)
    kind 'rows_user'  SASRec block rows [users, T, E]:                   (user * 32 + t) * E + c                           a4r_sasrec.hip:217,269
                      (sasrec_fused=False, the multi-launch user tower:   (user * T + t) * E + c, the GEMM epilogue's index on [users * T, E] rows)

Sites (adapter4rec_amd/engine.py): 999 embeddings, 16 i / 16 i + 1 / 16 i + 2 = probabilities / attention-output dense / FFN-output dense of
encoder layer i, 4000 user-encoder input, 4096 + 16 j (+ 1, + 2) the same three of SASRec block j.  The step's seed is
(engine.seed * 1000003 + engine.step_count) & 0xFFFFFFFFFFFF with step_count already advanced for the step (engine.py: train_forward).
"""
import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


def mix32(x):
    x = _u64(x) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash64(seed, site, idx):
    seed, site, idx = int(seed), int(site), _u64(idx)
    s0 = np.uint64((seed & 0xFFFFFFFF) ^ ((site * 0x9E3779B1) & 0xFFFFFFFF))
    s1 = np.uint64(((seed >> 32) + site * 0x85EBCA77) & 0xFFFFFFFF)
    c = (idx & _M32) ^ (((idx >> np.uint64(32)) * np.uint64(0xC2B2AE3D)) & _M32)
    lo = mix32((c * np.uint64(2) + s0) & _M32)
    hi = mix32((((c * np.uint64(2) + np.uint64(1)) & _M32) ^ s1 ^ lo))
    return (hi << np.uint64(32)) | lo


def thr16(p):
    if p <= 0:
        return 0
    t = np.float32(p) * np.float32(65536.0) + np.float32(0.5)
    return 65535 if t >= np.float32(65535.0) else int(t)


def keep_scale(p):
    return np.float32(1.0) / (np.float32(1.0) - np.float32(thr16(p)) / np.float32(65536.0)) if p > 0 else np.float32(1.0)


def keep(seed, site, e, p):
    e = _u64(e)
    h = hash64(seed, site, e >> np.uint64(2))
    lot = (h >> (np.uint64(16) * (e & np.uint64(3)))) & np.uint64(0xFFFF)
    return lot >= np.uint64(thr16(p))


class DropoutStream:
    """mask(kind, site, x, p, **geometry) -> float32 tensor of x's shape holding 0 or 1 / (1 - thr16 / 65536)"""

    def __init__(self, seed, sasrec_fused=True):
        self.seed = int(seed)
        self.sasrec_fused = bool(sasrec_fused)      # one launch per SASRec block (a4r_sasrec.hip: 32 rows per user) or the multi-launch user tower
        self.used = []                              # (GEMM epilogues on [users * T, E] rows: plain 'rows'); engine._sas_fused_ok() says which ran

    def mask(self, kind, site, x, p, head_dim=None):
        self.used.append((kind, int(site)))
        shp = tuple(x.shape)
        if p <= 0:
            return torch.ones(shp)
        if kind == 'rows_cls':                               # [items, S, H]: token 0 of every item as row `item` of an [items, H] tensor
            m = torch.ones(shp)
            m[:, 0, :] = self.mask('rows', site, x[:, 0, :], p)
            self.used.pop()
            return m
        if kind == 'rows':
            e = np.arange(int(np.prod(shp)), dtype=np.uint64)
        elif kind == 'rows_user':
            u, t, c = np.meshgrid(*[np.arange(n, dtype=np.uint64) for n in shp], indexing='ij')
            e = (u * np.uint64(32 if self.sasrec_fused else shp[1]) + t) * np.uint64(shp[2]) + c
        elif kind == 'attn_user':
            u, h, q, k = np.meshgrid(*[np.arange(n, dtype=np.uint64) for n in shp], indexing='ij')
            if shp[2] <= 32:
                e = (((u * np.uint64(shp[1]) + h) * np.uint64(32) + q) << np.uint64(5)) + k
            else:                                            # --max_seq_len above 32: the long attention kernels' index
                e = (((u * np.uint64(shp[1]) + h) * np.uint64(256) + q) << np.uint64(8)) + k
        elif kind == 'attn_item':
            i, h, q, k = np.meshgrid(*[np.arange(n, dtype=np.uint64) for n in shp], indexing='ij')
            pair, S = i * np.uint64(shp[1]) + h, shp[2]
            if head_dim is not None and head_dim <= 16:
                e = (pair * np.uint64(S) + q) * np.uint64(S) + k
            elif S <= 32:
                e = (pair * np.uint64(32) + q) * np.uint64(32) + k
            else:
                e = ((pair * np.uint64(256) + q) << np.uint64(8)) + k
        else:
            raise ValueError(kind)
        m = keep(self.seed, site, e.reshape(-1), p).reshape(shp)
        return torch.from_numpy(m.astype(np.float32) * keep_scale(p))
