"""Minimal LMDB 0.9 file writer for the tests (there is no liblmdb / py-lmdb in this image): builds a
data.mdb with two meta pages, leaf pages, overflow pages for large values and branch pages above them,
from sorted (key, value) byte pairs.  Written from the LMDB on-disk format description that
caffe_facade/src/lmdb_reader.cpp cites; it exercises the reader's tree walk, it does not prove
compatibility with liblmdb."""
import os
import struct

PSIZE = 4096
HDR = 16
MAGIC = 0xBEEFC0DE


def _page(pgno, flags, nodes):
    """nodes: list of bytes (each a complete node).  Nodes are packed from the end of the page."""
    buf = bytearray(PSIZE)
    upper = PSIZE
    ptrs = []
    for n in nodes:
        n = n + b"\0" * (len(n) & 1)          # nodes are 2-byte aligned
        upper -= len(n)
        buf[upper:upper + len(n)] = n
        ptrs.append(upper)
    lower = HDR + 2 * len(nodes)
    assert lower <= upper, "page overflow"
    struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
    for i, p in enumerate(ptrs):
        struct.pack_into("<H", buf, HDR + 2 * i, p)
    return bytes(buf)


def write_lmdb(directory, items):
    items = sorted(items)
    pages = {}                      # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p
    leaves, cur, cur_size = [], [], HDR
    n_overflow = 0

    def flush():
        nonlocal cur, cur_size
        if cur:
            pg = alloc()
            pages[pg] = _page(pg, 0x02, [n for _, n in cur])
            leaves.append((cur[0][0], pg))
            cur, cur_size = [], HDR
    for k, v in items:
        if 8 + len(k) + len(v) > PSIZE // 4:          # big value -> overflow pages
            npg = (HDR + len(v) + PSIZE - 1) // PSIZE
            ov = alloc(npg)
            blob = bytearray(npg * PSIZE)
            struct.pack_into("<QHHI", blob, 0, ov, 0, 0x04, npg)
            blob[HDR:HDR + len(v)] = v
            for i in range(npg):
                pages[ov + i] = bytes(blob[i * PSIZE:(i + 1) * PSIZE])
            n_overflow += npg
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0x01, len(k)) + k + struct.pack("<Q", ov)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        need = len(node) + (len(node) & 1) + 2
        if cur_size + need > PSIZE:
            flush()
        cur.append((k, node))
        cur_size += need
    flush()
    n_leaf, n_branch, depth = len(leaves), 0, 1 if leaves else 0
    level = leaves
    while len(level) > 1:            # build branch levels, up to 100 children per page
        nxt = []
        for i in range(0, len(level), 100):
            grp = level[i:i + 100]
            pg = alloc()
            nodes = []
            for j, (k, child) in enumerate(grp):
                kk = b"" if j == 0 else k
                nodes.append(struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, child >> 32, len(kk)) + kk)
            pages[pg] = _page(pg, 0x01, nodes)
            nxt.append((grp[0][0], pg))
            n_branch += 1
        level = nxt
        depth += 1
    root = level[0][1] if level else 0xFFFFFFFFFFFFFFFF
    last = next_pg[0] - 1

    def meta(pgno, txnid):
        buf = bytearray(PSIZE)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, 0x08, 0, 0)
        free_db = struct.pack("<IHHQQQQQ", PSIZE, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFFFFFFFFF)
        main_db = struct.pack("<IHHQQQQQ", 0, 0, depth, n_branch, n_leaf, n_overflow, len(items), root)
        m = struct.pack("<IIQQ", MAGIC, 1, 0, 1 << 30) + free_db + main_db + struct.pack("<QQ", last, txnid)
        buf[HDR:HDR + len(m)] = m
        return bytes(buf)
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "data.mdb"), "wb") as f:
        f.write(meta(0, 1))          # older snapshot (empty-ish): the reader must prefer txnid 2
        f.write(meta(1, 2))
        for pg in range(2, last + 1):
            f.write(pages[pg])
    return dict(leaf=n_leaf, branch=n_branch, overflow=n_overflow, depth=depth)
