"""Image input pipeline: record decode, Pillow-exact bilinear resize, uint8 -> normalised patches.
The checker for the resize is the installed Pillow itself (what torchvision.transforms.Resize calls on a PIL image);
oracle/pil_resize.py is its numpy restatement, pinned here bit-exactly."""
import pickle
import random
import sys
import types

import numpy as np
import pytest
import torch
from PIL import Image

import sim_lib
from oracle import pil_resize as P

SIZES = [(300, 200), (224, 224), (100, 640), (57, 33), (500, 375), (224, 300), (16, 16), (225, 223)]


def pil_resize(a, R):
    return np.asarray(Image.fromarray(a).convert('RGB').resize((R, R), Image.BILINEAR))


@pytest.mark.parametrize('hw', SIZES)
def test_oracle_resize_is_pillow(hw):
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    a = rng.integers(0, 256, (*hw, 3), dtype=np.uint8)
    for R in (224, 32):
        np.testing.assert_array_equal(P.resize_bilinear_u8(a, R), pil_resize(a, R))


def test_host_tables_match_oracle():
    from adapter4rec_amd.cv.image_io import resample_tables
    for n_in, n_out in ((300, 224), (100, 224), (224, 224), (33, 32), (1024, 224)):
        b, k = resample_tables(n_in, n_out, 'cpu')
        ob, ok = P.coeffs(n_in, n_out)
        np.testing.assert_array_equal(b.numpy(), ob)
        np.testing.assert_array_equal(k.numpy(), ok)


def test_decode_record_formats():
    from adapter4rec_amd.cv.image_io import LMDB_Image, RecordStore, decode_record
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    st = RecordStore()
    st.add(b'k1', a)
    np.testing.assert_array_equal(decode_record(st[b'k1']), a)
    # a record pickled where the reference pickles it: class data_utils.dataset.LMDB_Image (dataset.py:17-27)
    mod = types.ModuleType('data_utils.dataset')
    pkg = types.ModuleType('data_utils')
    cls = type('LMDB_Image', (), {'__init__': LMDB_Image.__init__, 'get_image': LMDB_Image.get_image, '__module__': 'data_utils.dataset'})
    mod.LMDB_Image = cls
    sys.modules['data_utils'], sys.modules['data_utils.dataset'] = pkg, mod
    try:
        blob = pickle.dumps(cls(a, 7))
    finally:
        del sys.modules['data_utils'], sys.modules['data_utils.dataset']
    np.testing.assert_array_equal(decode_record(blob), a)
    g = rng.integers(0, 256, (20, 10, 1), dtype=np.uint8)
    np.testing.assert_array_equal(decode_record(pickle.dumps(LMDB_Image(g, 0))), np.repeat(g, 3, 2))     # L -> RGB
    with pytest.raises(pickle.UnpicklingError):
        decode_record(pickle.dumps(print))


def _store(rng, n_items, sizes):
    from adapter4rec_amd.cv.image_io import RecordStore
    st, keys, raw = RecordStore(), {}, {}
    for i in range(1, n_items + 1):
        a = rng.integers(0, 256, (*sizes[i % len(sizes)], 3), dtype=np.uint8)
        keys[i] = f'item{i}'.encode()
        st.add(keys[i], a, i)
        raw[i] = a
    return st, keys, raw


def test_dataset_sampling_and_transform(monkeypatch):
    """Build_Lmdb_Dataset (product, resample kernel simulated on CPU) == the reference's __getitem__ semantics restated with
    Pillow + the same random stream (dataset.py:85-113)."""
    import adapter4rec_amd.cv.image_io as IO
    monkeypatch.setattr(IO, 'L', sim_lib)
    rng = np.random.default_rng(5)
    st, keys, raw = _store(rng, 30, [(40, 30), (32, 32), (50, 64)])
    u2seq = {0: [3, 9, 1, 22, 7], 1: list(range(10, 31))}
    ds = IO.Build_Lmdb_Dataset(u2seq, 30, 20, st, keys, 32, device='cpu')
    for u, seq in u2seq.items():
        random.seed(100 + u)
        sample, mask = ds[u]
        random.seed(100 + u)
        L_, pad = 21, 21 - len(seq)
        ref = np.zeros((L_, 2, 32, 32, 3), np.uint8)
        for i in range(len(seq) - 1):
            ref[pad + i, 0] = pil_resize(raw[seq[i]], 32)
            neg = random.randint(1, 30)
            while neg in seq:
                neg = random.randint(1, 30)
            ref[pad + i, 1] = pil_resize(raw[neg], 32)
        ref[pad + len(seq) - 1, 0] = pil_resize(raw[seq[-1]], 32)
        np.testing.assert_array_equal(sample.numpy(), ref)
        np.testing.assert_array_equal(mask.numpy(), np.array([0] * pad + [1] * (len(seq) - 1), np.float32))


@pytest.mark.parametrize('psize,n,cap,newest', [(4096, 0, None, 1), (4096, 1, None, 1), (4096, 700, None, 1), (4096, 700, None, 0),
                                                  (4096, 900, 5, 1), (512, 300, 3, 1), (16384, 50, None, 0)])
def test_lmdb_reader_point_lookups(tmp_path, psize, n, cap, newest):
    """cv/lmdb_reader.py against files written by tests/lmdb_writer.py (the format of liblmdb 0.9.x restated twice; no liblmdb in the image):
    every key back byte for byte -- small values inside leaf pages, large ones on overflow pages, trees of depth 1 .. 5 --, absent keys
    (before the first, between two, after the last, a prefix, an extension) -> None, the newest of the two meta pages wins."""
    from adapter4rec_amd.cv.lmdb_reader import LmdbReader
    from lmdb_writer import write_lmdb
    rng = np.random.default_rng(n + psize)
    recs = {}
    for i in range(n):
        size = int(rng.choice([0, 1, 7, 100, psize // 2 - 40, psize // 2, psize - 16, psize - 15, 3 * psize + 5, 40000]))
        recs[f'item{int(rng.integers(0, 10 ** 6))}'.encode('ascii')] = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    if n:
        recs[b'__len__'] = pickle.dumps(len(recs))
    path = str(tmp_path / 'image.lmdb')
    st = write_lmdb(path, recs, psize=psize, newest_meta=newest, max_leaf_nodes=cap)
    db = LmdbReader(path)
    assert db.stat()['entries'] == len(recs) and db.stat()['depth'] == st['depth'] and db.psize == psize
    if cap:
        assert st['depth'] >= 4
    with db.begin() as txn:
        for k, v in recs.items():
            assert txn.get(k) == v, k
        ks = sorted(recs)
        for k in [b'', b'\x00', b'zzzz'] + [k + b'0' for k in ks[:50]] + [k[:-1] for k in ks[:50]] + [k[:-1] + bytes([k[-1] + 1]) for k in ks[:50]]:
            assert txn.get(k) == recs.get(k), k
    db.close()


def test_lmdb_reader_refuses_what_it_does_not_read(tmp_path):
    from adapter4rec_amd.cv.lmdb_reader import LmdbFormatError, LmdbReader
    from lmdb_writer import write_lmdb
    path = str(tmp_path / 'data.mdb')
    write_lmdb(path, {b'a': b'1' * 10000, b'b': b'2'})
    assert LmdbReader(str(tmp_path)).get(b'b') == b'2'                # a directory: <dir>/data.mdb, as lmdb.open(subdir=True) reads
    again = pickle.loads(pickle.dumps(LmdbReader(path)))              # (what a spawned DataLoader worker receives)
    assert again.get(b'a') == b'1' * 10000 and again.stat()['entries'] == 2
    raw = bytearray(open(path, 'rb').read())
    for off, val, what in [(16, b'\xDE\xC0\xEF\xBF', 'magic'), (20, b'\x02\x00\x00\x00', 'version'), (10, b'\x02\x00', 'meta')]:
        bad = bytearray(raw)
        bad[off:off + len(val)] = val
        open(path, 'wb').write(bad)
        with pytest.raises(LmdbFormatError, match=what):
            LmdbReader(path)
    open(path, 'wb').write(raw[:3 * 4096])                              # a truncated copy
    with pytest.raises(LmdbFormatError, match='truncated'):
        LmdbReader(path)
    bad = bytearray(raw)
    bad[4096 + 16 + 24 + 48 + 4] = 0x08                                 # MDB_INTEGERKEY on the main database of the live meta page
    open(path, 'wb').write(bad)
    with pytest.raises(LmdbFormatError, match='integer keys'):
        LmdbReader(path)


def test_dataset_reads_an_lmdb_file(tmp_path, monkeypatch):
    """--lmdb_data pointing at an LMDB file in an image without the lmdb module: open_image_db -> LmdbReader -> Build_Lmdb_Dataset gives the
    samples the in-memory RecordStore gives (dataset.py:69-74,95-113)."""
    import adapter4rec_amd.cv.image_io as IO
    from adapter4rec_amd.cv.data_utils import open_image_db
    from lmdb_writer import write_lmdb
    monkeypatch.setattr(IO, 'L', sim_lib)
    monkeypatch.setitem(sys.modules, 'lmdb', None)                      # (import lmdb -> ImportError, also where the module exists)
    rng = np.random.default_rng(6)
    st, keys, raw = _store(rng, 30, [(40, 30), (32, 32), (50, 64)])
    path = str(tmp_path / 'image.lmdb')
    write_lmdb(path, dict(st))
    db = open_image_db(path)
    assert type(db).__name__ == 'LmdbReader'
    u2seq = {0: [3, 9, 1, 22, 7], 1: list(range(10, 31))}
    a, b = IO.Build_Lmdb_Dataset(u2seq, 30, 20, st, keys, 32, device='cpu'), IO.Build_Lmdb_Dataset(u2seq, 30, 20, db, keys, 32, device='cpu')
    for u in u2seq:
        random.seed(7 + u)
        sa, ma = a[u]
        random.seed(7 + u)
        sb, mb = b[u]
        assert torch.equal(sa, sb) and torch.equal(ma, mb)


@pytest.mark.gpu
@pytest.mark.parametrize('hw', SIZES)
def test_gpu_resize_is_pillow(hw):
    from adapter4rec_amd.cv.image_io import resize_to_square
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    a = rng.integers(0, 256, (3, *hw, 3), dtype=np.uint8)
    got = resize_to_square(torch.from_numpy(a).cuda(), 224).cpu().numpy()
    for j in range(3):
        np.testing.assert_array_equal(got[j], pil_resize(a[j], 224))


@pytest.mark.gpu
def test_gpu_raw_records_to_embeddings():
    """records of different sizes -> decode -> GPU resize -> uint8 patch path == the oracle on Pillow-resized, normalised floats."""
    import test_engine_cv as TC
    from adapter4rec_amd.cv.image_io import Build_Lmdb_Dataset
    from oracle import ref_cpu as R
    root, args, sd, cfg, fx, images, mask, noise = TC.build('cv_vit_houlsby', device='cuda:0')
    rng = np.random.default_rng(9)
    st, keys, raw = _store(rng, 40, [(40, 30), (32, 32), (50, 64), (20, 45)])
    ds = Build_Lmdb_Dataset({0: list(range(1, 22))}, 40, 20, st, keys, 32, device='cuda:0')
    random.seed(3)
    sample, m = ds[0]
    u8 = sample.view(-1, 32, 32, 3)
    emb = root.cv_encoder(u8).cpu()
    ref_in = torch.from_numpy(np.stack([((x.astype(np.float32) / 255 - 0.5) / 0.5).transpose(2, 0, 1) for x in u8.cpu().numpy()]))
    with torch.no_grad():
        ref = R.image_encoder(sd, ref_in, cfg)
    np.testing.assert_allclose(emb.numpy(), ref.numpy(), atol=1e-4, rtol=0)
    # and the resized bytes themselves are Pillow's
    random.seed(3)
    np.testing.assert_array_equal(sample[0, 0].cpu().numpy(), pil_resize(raw[1], 32))
