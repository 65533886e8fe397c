"""Datasets / samplers with the reference's interface (Downstream/Text/data_utils/dataset.py)."""
import math
import random

import numpy as np
import torch
from torch.utils.data import Dataset


class BuildTrainDataset(Dataset):
    """dataset.py:10-49: user -> (item contents [L, 2, 2*words] int64, log_mask [L-1] fp32); L = max_seq_len + 1.
    Left-pads to L, masks the pad positions, and for every real position draws ONE negative uniformly from
    1..item_num with python's `random`, rejecting items of the user's own sequence."""

    def __init__(self, u2seq, item_content, item_num, max_seq_len, use_modal):
        self.u2seq, self.item_content, self.item_num = u2seq, item_content, item_num
        self.max_seq_len, self.use_modal = max_seq_len + 1, use_modal
        # the gather below runs once per user and step on the host: an int64 copy indexed by a numpy array is 50x cheaper than the
        # reference's int32 matrix indexed by a tensor and converted per sample (100 + 10 us -> 2 us; same values, same dtype out)
        self._content64 = np.ascontiguousarray(np.asarray(item_content), dtype=np.int64) if use_modal else None

    def __len__(self):
        return len(self.u2seq)

    def __getitem__(self, user_id):
        seq = self.u2seq[user_id]
        pad = self.max_seq_len - len(seq)
        n_tok = len(seq) - 1
        log_mask = [0] * pad + [1] * n_tok
        negs = []
        for _ in range(n_tok):
            s = random.randint(1, self.item_num)
            while s in seq:
                s = random.randint(1, self.item_num)
            negs.append(s)
        ids = np.array([[0] * pad + list(seq), [0] * pad + negs + [0]], dtype=np.int64).T        # [L, 2]
        if self.use_modal:
            return torch.from_numpy(self._content64[ids]), torch.FloatTensor(log_mask)
        return torch.from_numpy(np.ascontiguousarray(ids)), torch.FloatTensor(log_mask)


class DeviceTrainSampler:
    """Device-side replacement of BuildTrainDataset + DataLoader for throughput runs ("a device-side sampler may
    replace it -- distribution pinned, not bit-pinned"): all user sequences live on the GPU as one left-padded [U, L] table, a batch
    is a gather + one vectorised rejection loop.  Same distribution as dataset.py:24-49: one negative per real position, uniform over
    1..item_num minus the user's own sequence; negatives row = [0]*pad + negs + [0]; log_mask = [0]*pad + [1]*(len - 1).
    ``sample(user_ids)`` -> (item_content[ids] [B*L*2, 2*words] int64 on the device, log_mask [B, L-1] fp32 on the HOST)."""

    def __init__(self, u2seq, item_content, item_num, max_seq_len, device, seed=0):
        self.L, self.item_num, self.device = max_seq_len + 1, item_num, torch.device(device)
        users = sorted(u2seq)
        tab = np.zeros((len(users), self.L), dtype=np.int64)
        for r, u in enumerate(users):
            seq = list(u2seq[u])
            tab[r, self.L - len(seq):] = seq
        self.row_of = {u: r for r, u in enumerate(users)}
        self.seqs = torch.from_numpy(tab).to(self.device)
        self.valid_host = tab != 0                 # (the pad structure of every user is known on the host: log_mask is handed over as a HOST tensor)
        self.content = torch.as_tensor(np.asarray(item_content)).long().to(self.device)
        self.seed = int(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(self.seed)

    def set_epoch(self, epoch):
        """Re-seed per epoch (a resumed run draws the negatives the uninterrupted one would have drawn in that epoch)."""
        self.gen.manual_seed(self.seed * 1000003 + int(epoch))

    def sample(self, user_ids):
        row_list = [self.row_of[int(u)] for u in user_ids]
        rows = torch.as_tensor(row_list, device=self.device)
        seq = self.seqs[rows]                                              # [B, L], 0 = pad
        valid = seq != 0
        vh = self.valid_host[row_list]
        # positions that have an input AND a target; computed from the host table and returned ON THE HOST (as the DataLoader's log_mask is): Model.forward
        # uploads it, and the engine finds the batch's pad slots in the host copy without a synchronisation (they are not encoded)
        log_mask = torch.from_numpy((vh[:, :-1] & vh[:, 1:]).astype(np.float32))
        need = torch.cat([valid[:, 1:], torch.zeros_like(valid[:, :1])], 1) & valid        # every real slot except the last one
        neg = torch.randint(1, self.item_num + 1, seq.shape, device=self.device, generator=self.gen)
        # rejection: redraw the slots that hit the user's own items.  A slot clashes with probability <= L / item_num per draw: with a real
        # catalogue (>= 4 096 items) FOUR unconditional redraw rounds leave < 1e-9 per slot and the host never waits for the device (a
        # `bool(clash.any())` per batch stalled the launch queue); tiny catalogues (tests) keep the checked loop
        checked = self.item_num < 4096
        for _ in range(64 if checked else 4):
            clash = (neg.unsqueeze(2) == seq.unsqueeze(1)).any(2) & need
            if checked and not bool(clash.any()):
                break
            neg = torch.where(clash, torch.randint(1, self.item_num + 1, seq.shape, device=self.device, generator=self.gen), neg)
        neg = torch.where(need, neg, torch.zeros_like(neg))
        ids = torch.stack([seq, neg], 2).reshape(-1)                       # [B, L, 2] -> rows (b, l, pos|neg)
        return self.content[ids].contiguous(), log_mask


class BuildEvalDataset(Dataset):
    """dataset.py:52-78 (kept for interface parity; the native eval path does not materialise one-hot labels)."""

    def __init__(self, u2seq, item_content, max_seq_len, item_num):
        self.u2seq, self.item_content, self.max_seq_len, self.item_num = u2seq, item_content, max_seq_len + 1, item_num

    def __len__(self):
        return len(self.u2seq)

    def __getitem__(self, user_id):
        seq = self.u2seq[user_id]
        tokens, target = seq[:-1], seq[-1]
        pad = self.max_seq_len - len(seq)
        log_mask = [0] * pad + [1] * len(tokens)
        labels = np.zeros(self.item_num)
        labels[target - 1] = 1.0
        return torch.LongTensor([user_id]), self.item_content[[0] * pad + tokens], torch.FloatTensor(log_mask), labels


class SequentialDistributedSampler(torch.utils.data.sampler.Sampler):
    """dataset.py:81-108: contiguous shards, tail padded with the last index up to a multiple of batch_size * world."""

    def __init__(self, dataset, batch_size, rank=None, num_replicas=None):
        import torch.distributed as dist
        self.num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
        self.rank = dist.get_rank() if rank is None else rank
        self.dataset, self.batch_size = dataset, batch_size
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / batch_size / self.num_replicas)) * batch_size
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        idx = list(range(len(self.dataset)))
        idx += [idx[-1]] * (self.total_size - len(idx))
        return iter(idx[self.rank * self.num_samples:(self.rank + 1) * self.num_samples])

    def indices(self):
        """this rank's indices as an int64 array (what __iter__ yields; eval_model takes them without a Python list of 10^5 ints)"""
        import numpy as np
        n = len(self.dataset)
        idx = np.minimum(np.arange(self.rank * self.num_samples, (self.rank + 1) * self.num_samples, dtype=np.int64), n - 1)
        return idx

    def __len__(self):
        return self.num_samples
