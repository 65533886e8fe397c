"""Image input pipeline of the reference's CV path (Downstream/CV/data_utils/dataset.py:17-27,60-113), MI355X-side:

    LMDB value bytes --decode_record--> uint8 [H, W, C]  --H2D (raw bytes, 4x fewer than fp32)-->
    resize_to_square (two a4r_resample_u8 passes: Pillow's bilinear, bit-exact) --> uint8 [n, R, R, 3]
    --> the engine's patch kernel applies ToTensor + Normalize(0.5, 0.5) while it builds the patch matrix.

The reference does all of it per sample on CPU workers (PIL + torchvision) and ships 25 MB of fp32 per user.
``lmdb`` itself is not in this image: ``RecordStore`` is the same key -> pickled-``LMDB_Image`` mapping held in memory (or
backed by any object with ``get(key) -> bytes``, e.g. an ``lmdb`` transaction where the module exists)."""
import io
import math
import pickle
import random

import numpy as np
import torch

from .. import _lib as L

PRECISION_BITS = 32 - 8 - 2


class LMDB_Image:                                 # dataset.py:17-27 (the pickled record type)
    def __init__(self, image, id):
        self.channels = image.shape[2]
        self.size = image.shape[:2]
        self.image = image.tobytes()
        self.id = id

    def get_image(self):
        return np.frombuffer(self.image, dtype=np.uint8).reshape(*self.size, self.channels)


class _RecordUnpickler(pickle.Unpickler):
    """Records were pickled as ``data_utils.dataset.LMDB_Image`` (or ``__main__.LMDB_Image``): map any such name to the class
    above and refuse everything that is not plain data."""
    SAFE = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'), ('numpy', 'dtype')}

    def find_class(self, module, name):
        if name == 'LMDB_Image':
            return LMDB_Image
        if (module, name) in self.SAFE:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'unexpected global {module}.{name} in an image record')


def decode_record(value):
    """LMDB value (``pickle.dumps(LMDB_Image(...))``) -> uint8 [H, W, 3] (``Image.fromarray(...).convert('RGB')``)."""
    rec = _RecordUnpickler(io.BytesIO(value)).load()
    img = rec.get_image()
    if img.ndim == 2 or img.shape[2] == 1:
        img = np.repeat(img.reshape(img.shape[0], img.shape[1], 1), 3, axis=2)
    elif img.shape[2] == 4:
        img = img[:, :, :3]                       # RGBA -> RGB drops alpha (PIL convert)
    return np.ascontiguousarray(img)


class RecordStore(dict):
    """key (bytes) -> record bytes; ``add(key, uint8 image)`` writes what the reference's LMDB builder writes."""

    def add(self, key, image, id=0):
        self[key] = pickle.dumps(LMDB_Image(np.ascontiguousarray(image), id))


_TABLES = {}


def resample_tables(in_size, out_size, device):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter -> device int32 (bounds, kk)."""
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        scale = filterscale = in_size / out_size
        if filterscale < 1.0:
            filterscale = 1.0
        support = filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        bounds = np.zeros((out_size, 2), np.int32)
        kk = np.zeros((out_size, ksize), np.int32)
        ss = 1.0 / filterscale
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            cnt = min(int(center + support + 0.5), in_size) - xmin
            w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(cnt)]
            ww = 0.0
            for v in w:
                ww += v
            bounds[xx] = (xmin, cnt)
            for x, v in enumerate(w):
                kk[xx, x] = int(0.5 + (v / ww if ww != 0.0 else v) * (1 << PRECISION_BITS))
        _TABLES[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device))
    return _TABLES[key]


def resize_to_square(images_u8, R):
    """uint8 [n, H, W, C] on the GPU -> uint8 [n, R, R, C]; horizontal pass, 8-bit intermediate, vertical pass (Pillow's order)."""
    L.require_gpu(images_u8)
    n, H, W, C = images_u8.shape
    x = images_u8.contiguous()
    if W != R:
        b, k = resample_tables(W, R, x.device)
        y = torch.empty(n, H, R, C, dtype=torch.uint8, device=x.device)
        L.resample_u8(x, y, b, k, n * H, W, R, C)
        x = y
    if H != R:
        b, k = resample_tables(H, R, x.device)
        y = torch.empty(n, R, R, C, dtype=torch.uint8, device=x.device)
        L.resample_u8(x, y, b, k, n, H, R, R * C)
        x = y
    return x


class Build_Lmdb_Dataset(torch.utils.data.Dataset):
    """Drop-in for dataset.py:60-113 with the CPU transform removed: yields (uint8 [L, 2, R, R, 3] on ``device``, log_mask).
    Sampling is the reference's: left-padded slots, one ``random.randint`` negative per position rejected while in the
    user's sequence, last negative slot and pad slots left empty (zeros)."""

    def __init__(self, u2seq, item_num, max_seq_len, db, item_id_to_keys, resize, device='cuda', host=False):
        """host=True (DataLoader worker processes, --num_workers > 0): __getitem__ touches no device -- it returns the decoded records stacked by
        source size + their slots; collate_host() merges a batch's groups and assemble_batch() uploads / resizes / scatters them in the training process"""
        self.u2seq, self.item_num, self.max_seq_len = u2seq, item_num, max_seq_len + 1
        self.db, self.item_id_to_keys, self.resize, self.device = db, item_id_to_keys, resize, torch.device(device)
        self.host = bool(host)

    def __len__(self):
        return len(self.u2seq)

    def _load(self, item_ids):
        """decode + group by source size + one H2D copy and two resample launches per group."""
        raw = [decode_record(self.db.get(self.item_id_to_keys[i])) for i in item_ids]
        out = torch.zeros(len(raw), self.resize, self.resize, 3, dtype=torch.uint8, device=self.device)
        groups = {}
        for j, a in enumerate(raw):
            groups.setdefault(a.shape, []).append(j)
        for shape, idx in groups.items():
            batch = torch.from_numpy(np.stack([raw[j] for j in idx])).to(self.device, non_blocking=True)
            out[torch.tensor(idx, device=self.device)] = resize_to_square(batch, self.resize)
        return out

    def __getitem__(self, user_id):
        seq = self.u2seq[user_id]
        seq_len, tokens_len = len(seq), len(seq) - 1
        mask_len = self.max_seq_len - seq_len
        log_mask = [0] * mask_len + [1] * tokens_len
        ids, slots = [], []
        for i in range(tokens_len):
            ids.append(seq[i]); slots.append((mask_len + i, 0))
            neg = random.randint(1, self.item_num)
            while neg in seq:
                neg = random.randint(1, self.item_num)
            ids.append(neg); slots.append((mask_len + i, 1))
        ids.append(seq[-1]); slots.append((mask_len + tokens_len, 0))
        if self.host:
            raw = [decode_record(self.db.get(self.item_id_to_keys[i])) for i in ids]
            groups = {}
            for j, a in enumerate(raw):
                groups.setdefault(tuple(a.shape), []).append(j)
            out = []
            for shape, idx in groups.items():
                out.append((torch.from_numpy(np.stack([raw[j] for j in idx])), torch.tensor([slots[j][0] for j in idx]), torch.tensor([slots[j][1] for j in idx])))
            return out, torch.tensor(log_mask, dtype=torch.float32)
        imgs = self._load(ids)
        sample = torch.zeros(self.max_seq_len, 2, self.resize, self.resize, 3, dtype=torch.uint8, device=self.device)
        rows = torch.tensor([s[0] for s in slots], device=self.device)
        cols = torch.tensor([s[1] for s in slots], device=self.device)
        sample[rows, cols] = imgs
        return sample, torch.tensor(log_mask, dtype=torch.float32)


def collate_host(batch):
    """DataLoader collate_fn for Build_Lmdb_Dataset(host=True): the batch's records merged per source size
    -> ([(uint8 [n, H, W, 3], sample index [n], slot row [n], slot column [n]) per size], log_mask [B, L - 1]); plain tensors, so pin_memory applies"""
    by_shape = {}
    for b, (groups, _) in enumerate(batch):
        for t, rows, cols in groups:
            g = by_shape.setdefault(tuple(t.shape[1:]), ([], [], [], []))
            g[0].append(t); g[1].append(torch.full((t.shape[0],), b, dtype=torch.long)); g[2].append(rows); g[3].append(cols)
    merged = [(torch.cat(g[0]), torch.cat(g[1]), torch.cat(g[2]), torch.cat(g[3])) for g in by_shape.values()]
    return merged, torch.stack([m for _, m in batch])


def assemble_batch(merged, B, L, R, device):
    """the training process's half of the host path: one H2D copy + two resample launches per source size, scattered into uint8 [B, L, 2, R, R, 3]
    (pad slots and every user's last negative stay zero, as Build_Lmdb_Dataset.__getitem__ leaves them)"""
    out = torch.zeros(B, L, 2, R, R, 3, dtype=torch.uint8, device=device)
    for t, bidx, rows, cols in merged:
        x = t.to(device, non_blocking=True)
        y = resize_to_square(x, R) if (x.shape[1] != R or x.shape[2] != R) else x
        out[bidx.to(device, non_blocking=True), rows.to(device, non_blocking=True), cols.to(device, non_blocking=True)] = y
    return out
