"""Test-side writer of LMDB data files (format version 1, liblmdb 0.9.x layout: mdb.c MDB_meta / MDB_db / MDB_page / MDB_node), so that
adapter4rec_amd/cv/lmdb_reader.py has files to open in an image without the ``lmdb`` module.  A second, independent restatement of the
same published format (bulk load of sorted keys: leaf pages filled bottom-up, values above liblmdb's node limit on overflow pages, branch
levels up to one root, both meta pages) -- it pins the reader to the format description, not to liblmdb's own output."""
import struct

MAGIC, VERSION, HDR = 0xBEEFC0DE, 1, 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1
P_INVALID = (1 << 64) - 1


def _even(n):
    return (n + 1) & ~1


def _page(psize, pgno, flags, nodes):
    """nodes: list of packed node bytes; offsets grow from byte 16, node bodies are placed from the end of the page downwards"""
    buf = bytearray(psize)
    upper = psize
    ptrs = []
    for nd in nodes:
        upper -= _even(len(nd))
        buf[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = HDR + 2 * len(nodes)
    assert lower <= upper, 'page overfull'
    struct.pack_into('<QHHHH', buf, 0, pgno, 0, flags, lower, upper)
    for i, p in enumerate(ptrs):
        struct.pack_into('<H', buf, HDR + 2 * i, p)
    return bytes(buf)


def _fits(psize, nodes, nd):
    used = HDR + sum(2 + _even(len(x)) for x in nodes)
    return used + 2 + _even(len(nd)) <= psize


def write_lmdb(path, records, psize=4096, newest_meta=1, max_leaf_nodes=None):
    """records: dict bytes -> bytes.  newest_meta: which of the two meta pages carries the committed tree (the other one an older, empty
    snapshot).  max_leaf_nodes: cap the nodes per page to force a deeper tree out of few records."""
    keys = sorted(records)
    nodemax = (((psize - HDR) // 2) & ~1) - 2
    pages = {}                                   # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    n_overflow = 0
    level = []                                   # (first key, pgno) of the pages of the current level
    cur, cur_first = [], None

    def flush_leaf():
        nonlocal cur, cur_first
        if cur:
            pg = alloc()
            pages[pg] = _page(psize, pg, P_LEAF, cur)
            level.append((cur_first, pg))
        cur, cur_first = [], None
    for k in keys:
        v = records[k]
        if 8 + len(k) + len(v) > nodemax:
            n = (HDR + len(v) + psize - 1) // psize
            opg = alloc(n)
            buf = bytearray(n * psize)
            struct.pack_into('<QHHI', buf, 0, opg, 0, P_OVERFLOW, n)
            buf[HDR:HDR + len(v)] = v
            for j in range(n):
                pages[opg + j] = bytes(buf[j * psize:(j + 1) * psize])
            n_overflow += n
            nd = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack('<Q', opg)
        else:
            nd = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        if cur and (not _fits(psize, cur, nd) or (max_leaf_nodes and len(cur) >= max_leaf_nodes)):
            flush_leaf()
        if not cur:
            cur_first = k
        cur.append(nd)
    flush_leaf()
    n_leaf, n_branch, depth = len(level), 0, 1 if level else 0
    while len(level) > 1:
        nxt, cur, cur_first = [], [], None
        for first, pg in level:
            kk = b'' if not cur else first                  # node 0 of a branch page carries no key
            nd = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, pg >> 32, len(kk)) + kk
            if cur and (not _fits(psize, cur, nd) or (max_leaf_nodes and len(cur) >= max_leaf_nodes)):
                bp = alloc()
                pages[bp] = _page(psize, bp, P_BRANCH, cur)
                nxt.append((cur_first, bp))
                cur, cur_first = [], None
                nd = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, pg >> 32, 0)
            if not cur:
                cur_first = first
            cur.append(nd)
        bp = alloc()
        pages[bp] = _page(psize, bp, P_BRANCH, cur)
        nxt.append((cur_first, bp))
        n_branch += len(nxt)
        level, depth = nxt, depth + 1
        cur, cur_first = [], None
    root = level[0][1] if level else P_INVALID
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, live):
        buf = bytearray(psize)
        struct.pack_into('<QHHHH', buf, 0, pgno, 0, P_META, 0, 0)
        o = HDR
        struct.pack_into('<IIQQ', buf, o, MAGIC, VERSION, 0, 1 << 30)
        o += 24
        struct.pack_into('<IHHQQQQQ', buf, o, psize, 0, 0, 0, 0, 0, 0, P_INVALID)              # free-list database
        o += 48
        if live:
            struct.pack_into('<IHHQQQQQ', buf, o, 0, 0, depth, n_branch, n_leaf, n_overflow, len(keys), root)
        else:
            struct.pack_into('<IHHQQQQQ', buf, o, 0, 0, 0, 0, 0, 0, 0, P_INVALID)
        o += 48
        struct.pack_into('<QQ', buf, o, last_pg if live else 1, txnid)
        return bytes(buf)
    with open(path, 'wb') as f:
        f.write(meta(0, 2 if newest_meta == 0 else 0, newest_meta == 0))
        f.write(meta(1, 1, newest_meta == 1))
        for pg in range(2, next_pg[0]):
            f.write(pages[pg])
    return dict(depth=depth, leaf_pages=n_leaf, branch_pages=n_branch, overflow_pages=n_overflow, last_pgno=last_pg)
