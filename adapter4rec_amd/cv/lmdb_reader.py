"""Read-only access to an LMDB data file without the ``lmdb`` module -- what ``Build_Lmdb_Dataset`` needs of it
(Downstream/CV/data_utils/dataset.py:69-74,95-113, metrics.py:77-80: ``lmdb.open(path, subdir=isdir(path), readonly=True, lock=False,
readahead=False, meminit=False)``, ``env.begin()``, ``txn.get(key)`` on the unnamed database with byte-string keys).

The reference's algorithm here lives in a third-party dependency that is not in this image: py-lmdb over liblmdb (OpenLDAP LMDB 0.9.x,
data format version 1).  This module restates the published on-disk format of that version (mdb.c: MDB_meta / MDB_db / MDB_page /
MDB_node) for the one operation the path uses -- a point lookup in the main B+tree of the newest committed meta page:

  file    = pages of `psize` bytes; pages 0 and 1 are meta pages, the one with the larger txnid is current
  page    = header {pgno u64, pad u16, flags u16, lower u16, upper u16 | n_pages u32 (overflow)} + u16 node offsets from byte 16
  meta    = header + {magic u32 = 0xBEEFC0DE, version u32 = 1, address u64, mapsize u64, db[2] (free list, main), last_pg u64, txnid u64}
  db      = {pad u32 (db[0]: the page size), flags u16, depth u16, branch / leaf / overflow pages u64 x 3, entries u64, root u64}
  node    = {lo u16, hi u16, flags u16, ksize u16, key bytes, data}; branch: child page = lo | hi << 16 | flags << 32, node 0's key
            is implicit; leaf: data size = lo | hi << 16, flag 0x01 (F_BIGDATA) = the data is a u64 page number whose overflow
            page(s) hold the value from byte 16 on
  keys    = compared as byte strings (memcmp, then length): the default comparator, which is what the reference's ascii keys use

Only little-endian 64-bit files (the only kind liblmdb writes on the machines the reference runs on).  Named sub-databases, duplicate-sort
databases (F_SUBDATA / F_DUPDATA) and MDB_INTEGERKEY / MDB_REVERSEKEY main databases are refused loudly.

Parity note: no file written by liblmdb exists in this image, so the reader is checked against files produced by tests/lmdb_writer.py, a
second restatement of the same format (leaf, branch, overflow pages, two meta pages) -- "pinned to the format description, not to liblmdb's output".
Where the ``lmdb`` module is installed, ``open_image_db`` uses it and this file is not involved."""
import mmap
import os
import struct

MAGIC, DATA_VERSION = 0xBEEFC0DE, 1
PAGEHDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
MDB_REVERSEKEY, MDB_DUPSORT, MDB_INTEGERKEY = 0x02, 0x04, 0x08
P_INVALID = (1 << 64) - 1
_META = struct.Struct('<IIQQ')                  # magic, version, address, mapsize
_DB = struct.Struct('<IHHQQQQQ')                # pad, flags, depth, branch, leaf, overflow, entries, root
_NODE = struct.Struct('<HHHH')                  # lo, hi, flags, ksize


class LmdbFormatError(RuntimeError):
    pass


class LmdbReader:
    """``LmdbReader(path).get(key) -> bytes | None``; also ``begin()`` as a context manager returning itself, so that code written for
    ``env.begin() as txn: txn.get(key)`` runs unchanged.  The snapshot is the newest meta page at open time (a read-only transaction)."""

    def __init__(self, path, subdir=None):
        subdir = os.path.isdir(path) if subdir is None else subdir
        self.path = os.path.join(path, 'data.mdb') if subdir else path
        self._open()

    def __getstate__(self):                      # (a DataLoader worker started by spawn re-opens the file; the mapping itself does not pickle)
        return {'path': self.path}

    def __setstate__(self, st):
        self.path = st['path']
        self._open()

    def _open(self):
        self._f = open(self.path, 'rb')
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise LmdbFormatError(f'{self.path}: {size} bytes is too short for two meta pages')
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        m0 = self._meta_at(0)
        self.psize = m0['psize']
        if self.psize < 512 or self.psize & (self.psize - 1) or size < 2 * self.psize:
            raise LmdbFormatError(f'{self.path}: page size {self.psize}')
        m1 = self._meta_at(self.psize)
        meta = m1 if m1['txnid'] > m0['txnid'] else m0
        self.txnid, self.last_pg = meta['txnid'], meta['last_pg']
        self.flags, self.depth, self.entries, self.root = meta['flags'], meta['depth'], meta['entries'], meta['root']
        if self.flags & (MDB_REVERSEKEY | MDB_DUPSORT | MDB_INTEGERKEY):
            raise LmdbFormatError(f'{self.path}: main database flags {self.flags:#x} (reverse / duplicate-sort / integer keys) are not supported')
        if (self.last_pg + 1) * self.psize > size:
            raise LmdbFormatError(f'{self.path}: the meta page names page {self.last_pg} but the file holds {size // self.psize} pages (truncated copy?)')

    def _meta_at(self, off):
        pgno, _pad, flags, _lo, _up = struct.unpack_from('<QHHHH', self._m, off)
        if not flags & P_META:
            raise LmdbFormatError(f'{self.path}: page at byte {off} is not a meta page (flags {flags:#x}) -- not an LMDB data file')
        magic, version, _addr, _mapsize = _META.unpack_from(self._m, off + PAGEHDR)
        if magic != MAGIC:
            raise LmdbFormatError(f'{self.path}: magic {magic:#x}, expected {MAGIC:#x} -- not an LMDB data file (or a big-endian one)')
        if version != DATA_VERSION:
            raise LmdbFormatError(f'{self.path}: LMDB data format version {version}; this reader knows version {DATA_VERSION} (liblmdb 0.9.x)')
        o = off + PAGEHDR + _META.size
        free = _DB.unpack_from(self._m, o)
        main = _DB.unpack_from(self._m, o + _DB.size)
        last_pg, txnid = struct.unpack_from('<QQ', self._m, o + 2 * _DB.size)
        return dict(psize=free[0], flags=main[1], depth=main[2], entries=main[6], root=main[7], last_pg=last_pg, txnid=txnid)

    # -- pages ---------------------------------------------------------------------------------------------------------------
    def _page(self, pgno):
        if pgno > self.last_pg:
            raise LmdbFormatError(f'{self.path}: page {pgno} beyond the last page {self.last_pg}')
        off = pgno * self.psize
        no, _pad, flags, lower, upper = struct.unpack_from('<QHHHH', self._m, off)
        if no != pgno:
            raise LmdbFormatError(f'{self.path}: page {pgno} carries number {no}')
        return off, flags, lower, upper

    def _node(self, off, i):
        ptr = struct.unpack_from('<H', self._m, off + PAGEHDR + 2 * i)[0]
        lo, hi, flags, ksize = _NODE.unpack_from(self._m, off + ptr)
        k0 = off + ptr + _NODE.size
        return lo, hi, flags, ksize, k0

    def _key(self, off, i):
        _lo, _hi, _fl, ksize, k0 = self._node(off, i)
        return self._m[k0:k0 + ksize]

    def _search(self, off, n, key, first):
        """smallest i in [first, n) with key_i >= key (mdb_node_search); (i, exact)"""
        lo, hi = first, n
        while lo < hi:
            mid = (lo + hi) // 2
            if self._key(off, mid) < key:
                lo = mid + 1
            else:
                hi = mid
        return lo, lo < n and self._key(off, lo) == key

    # -- the one operation ---------------------------------------------------------------------------------------------------
    def get(self, key, default=None):
        key = bytes(key)
        if self.root == P_INVALID or not key:
            return default
        pgno = self.root
        for _ in range(64):                                           # (a tree deeper than this is a cycle in a damaged file)
            off, flags, lower, _upper = self._page(pgno)
            n = (lower - PAGEHDR) >> 1
            if flags & P_LEAF2:
                raise LmdbFormatError(f'{self.path}: fixed-size-key leaf (MDB_DUPFIXED) in the main tree')
            if flags & P_BRANCH:
                i, exact = self._search(off, n, key, 1)               # node 0 of a branch page has no key: it covers everything below key_1
                if not exact:
                    i -= 1
                lo, hi, fl, _ks, _k0 = self._node(off, i)
                pgno = lo | hi << 16 | fl << 32
                continue
            if not flags & P_LEAF:
                raise LmdbFormatError(f'{self.path}: page {pgno} has flags {flags:#x} inside the tree')
            i, exact = self._search(off, n, key, 0)
            if not exact:
                return default
            lo, hi, fl, ksize, k0 = self._node(off, i)
            dsize = lo | hi << 16
            if fl & (F_SUBDATA | F_DUPDATA):
                raise LmdbFormatError(f'{self.path}: key {key!r} is a named sub-database or carries duplicates (node flags {fl:#x})')
            d0 = k0 + ksize
            if not fl & F_BIGDATA:
                return self._m[d0:d0 + dsize]
            opg = struct.unpack_from('<Q', self._m, d0)[0]
            ooff, oflags, _l, _u = self._page(opg)
            n_pages = struct.unpack_from('<I', self._m, ooff + 12)[0]
            if not oflags & P_OVERFLOW or PAGEHDR + dsize > n_pages * self.psize or opg + n_pages - 1 > self.last_pg:
                raise LmdbFormatError(f'{self.path}: value of {key!r}: {dsize} bytes in {n_pages} overflow page(s) at {opg} (flags {oflags:#x})')
            return self._m[ooff + PAGEHDR:ooff + PAGEHDR + dsize]
        raise LmdbFormatError(f'{self.path}: no leaf within 64 levels of the root')

    def stat(self):
        return dict(psize=self.psize, depth=self.depth, entries=self.entries, last_pgno=self.last_pg, txnid=self.txnid)

    # (py-lmdb's shape, so that `with env.begin() as txn: txn.get(key)` reads the same)
    def begin(self, *a, **k):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        self._m.close()
        self._f.close()
