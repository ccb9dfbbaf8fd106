"""Independent (test-side) writer/reader of ivf index pages in the reference's on-disk format
(PostgreSQL page layout + src/index/ivf_am.c structs), used to check neurondb_amd's page codec.
Nothing here is shared with the product code."""
import struct

import numpy as np

BLCKSZ = 8192
INVALID = 0xFFFFFFFF


def maxalign(x):
    return (x + 7) & ~7


class Page:
    def __init__(self, special):
        self.b = bytearray(BLCKSZ)
        self.lower = 24
        self.special = BLCKSZ - maxalign(special)
        self.upper = self.special
        self._hdr()

    def _hdr(self):
        struct.pack_into("<HHHH", self.b, 12, self.lower, self.upper, self.special, BLCKSZ | 4)

    def free(self):
        sp = self.upper - self.lower
        return 0 if sp < 4 else sp - 4

    def add(self, data, flags=1):
        size = len(data)
        self.upper -= maxalign(size)
        self.b[self.upper:self.upper + size] = data
        struct.pack_into("<I", self.b, self.lower, self.upper | (flags << 15) | (size << 17))
        self.lower += 4
        self._hdr()
        return (self.lower - 24) // 4          # offset number (1-based)

    def set_flags(self, offnum, flags):
        pos = 24 + 4 * (offnum - 1)
        w, = struct.unpack_from("<I", self.b, pos)
        struct.pack_into("<I", self.b, pos, (w & ~(3 << 15)) | (flags << 15))


def write_reference_format(centroids, lists, nprobe=10, dead=(), foreign_dim=()):
    """Version-1 image exactly as ivfbuild + a sequence of ivfinsert calls leave it:
    block 0 meta, block 1 all centroids, list pages allocated in insertion order.
    lists: per list, an array of (row vector, tid6 bytes). dead / foreign_dim: sets of (list, index)."""
    centroids = np.asarray(centroids, np.float32)
    nl, dim = centroids.shape
    pages = [Page(32), Page(24)]
    cent_off = []
    for c in range(nl):
        item = struct.pack("<iiqI", c, dim, 0, INVALID) + b"\0\0\0\0" + centroids[c].tobytes()
        assert pages[1].free() >= len(item), "reference format: centroids must fit ONE page (quirk Q6)"
        cent_off.append(pages[1].add(item))
    tails = [None] * nl
    total = 0
    esize = 8 + maxalign(4 * dim)
    # interleave inserts across lists round-robin to scatter the chains over the relation
    cursors = [0] * nl
    remaining = sum(len(l) for l in lists)
    while remaining:
        for L in range(nl):
            if cursors[L] >= len(lists[L]):
                continue
            vec, tid = lists[L][cursors[L]]
            if tails[L] is None or pages[tails[L]].free() < esize:
                pages.append(Page(8))
                nb = len(pages) - 1
                struct.pack_into("<Ii", pages[nb].b, pages[nb].special, INVALID, 0)
                if tails[L] is None:
                    # patch centroid->firstBlock
                    cp = pages[1]
                    w, = struct.unpack_from("<I", cp.b, 24 + 4 * (cent_off[L] - 1))
                    struct.pack_into("<I", cp.b, (w & 0x7FFF) + 16, nb)
                else:
                    struct.pack_into("<I", pages[tails[L]].b, pages[tails[L]].special, nb)
                tails[L] = nb
            d = dim + 1 if (L, cursors[L]) in foreign_dim else dim
            item = bytes(tid) + struct.pack("<h", d) + np.asarray(vec, np.float32).tobytes()
            item += b"\0" * (esize - len(item))
            off = pages[tails[L]].add(item)
            cnt, = struct.unpack_from("<i", pages[tails[L]].b, pages[tails[L]].special + 4)
            struct.pack_into("<i", pages[tails[L]].b, pages[tails[L]].special + 4, cnt + 1)
            if (L, cursors[L]) in dead:
                pages[tails[L]].set_flags(off, 3)          # LP_DEAD
            cursors[L] += 1
            remaining -= 1
            total += 1
    struct.pack_into("<IIiiiIq", pages[0].b, 24, 0x49564646, 1, nl, nprobe, dim, 1, total)
    return b"".join(bytes(p.b) for p in pages)


def read_image(img):
    """Independent reader (both versions): returns dim, centroids, list_len, rows, tids6."""
    nb = len(img) // BLCKSZ
    magic, version, nlists, nprobe, dim, cblock, inserted = struct.unpack_from("<IIiiiIq", img, 24)
    assert magic == 0x49564646
    cents, first = [], []
    cb = cblock
    while cb not in (0, INVALID):
        base = cb * BLCKSZ
        lower, upper, special, _ = struct.unpack_from("<HHHH", img, base + 12)
        for i in range((lower - 24) // 4):
            w, = struct.unpack_from("<I", img, base + 24 + 4 * i)
            off = w & 0x7FFF
            first.append(struct.unpack_from("<I", img, base + off + 16)[0])
            cents.append(np.frombuffer(img, np.float32, dim, base + off + 24))
        cb = struct.unpack_from("<I", img, base + special)[0] if version >= 2 else INVALID
    rows, tids, lens = [], [], []
    for fb in first:
        n = 0
        lb = fb
        while lb != INVALID:
            base = lb * BLCKSZ
            lower, upper, special, _ = struct.unpack_from("<HHHH", img, base + 12)
            for i in range((lower - 24) // 4):
                w, = struct.unpack_from("<I", img, base + 24 + 4 * i)
                off, flags = w & 0x7FFF, (w >> 15) & 3
                if flags != 1:
                    continue
                if struct.unpack_from("<h", img, base + off + 6)[0] != dim:
                    continue
                tids.append(img[base + off: base + off + 6])
                rows.append(np.frombuffer(img, np.float32, dim, base + off + 8))
                n += 1
            lb = struct.unpack_from("<I", img, base + special)[0]
        lens.append(n)
    rows = np.array(rows, np.float32).reshape(-1, dim)
    t6 = np.frombuffer(b"".join(tids), np.uint8).reshape(-1, 6) if tids else np.zeros((0, 6), np.uint8)
    return dim, np.array(cents, np.float32), np.array(lens, np.int64), rows, t6, version


# ---------------------------------------------------------------------------------------------
# hnsw relation (src/index/hnsw_am.c:108-181): block 0 meta, every other block one node item
# ---------------------------------------------------------------------------------------------

def write_hnsw_reference_format(vecs, levels, ncount, nbrs, tids6, entry_point, entry_level, m, efc=200, efs=64,
                                dead=()):
    """Image as hnswbuild/hnswinsert leave it (node b on block b, neighbour slots of levels 0..level only);
    `dead`: blocks whose line pointer hnswbulkdelete marked dead."""
    nb, dim = vecs.shape
    pages = [Page(40)]
    live = 0
    for b in range(1, nb):
        lv = int(levels[b])
        item = bytes(tids6[b]) + b"\0\0" + struct.pack("<ih", lv, dim) + np.asarray(ncount[b], np.int16).tobytes()
        assert len(item) == 46
        item += b"\0\0" + np.asarray(vecs[b], np.float32).tobytes()
        item += np.asarray(nbrs[b, :lv + 1], np.uint32).tobytes()
        pg = Page(0)
        off = pg.add(item)
        if b in dead:
            pg.set_flags(off, 3)
        else:
            live += 1
        pages.append(pg)
    struct.pack_into("<IIIiihhhxxfq", pages[0].b, 24, 0x48534E57, 1, entry_point & 0xFFFFFFFF, entry_level,
                     int(max(levels[1:])) if nb > 1 else -1, m, efc, efs, 0.36, live)
    return b"".join(bytes(p.b) for p in pages)


def read_hnsw_image(img, m):
    """Independent reader -> dense arrays (slots above a node's level = 0xFFFFFFFF) + dead flags + meta."""
    nb = len(img) // BLCKSZ
    magic, version, entry, entry_level, max_level, mm, efc, efs, ml, inserted = \
        struct.unpack_from("<IIIiihhhxxfq", img, 24)
    assert magic == 0x48534E57 and mm == m
    vecs = levels = None
    ncount = np.zeros((nb, 16), np.int16)
    nbrs = np.full((nb, 16, 2 * m), INVALID, np.uint32)
    tids = np.zeros((nb, 6), np.uint8)
    dead = np.zeros(nb, np.uint8)
    levels = np.zeros(nb, np.int32)
    for b in range(1, nb):
        base = b * BLCKSZ
        w, = struct.unpack_from("<I", img, base + 24)
        off, flags = w & 0x7FFF, (w >> 15) & 3
        lv, dim = struct.unpack_from("<ih", img, base + off + 8)
        if vecs is None:
            vecs = np.zeros((nb, dim), np.float32)
        tids[b] = np.frombuffer(img, np.uint8, 6, base + off)
        levels[b] = lv
        ncount[b] = np.frombuffer(img, np.int16, 16, base + off + 14)
        vecs[b] = np.frombuffer(img, np.float32, dim, base + off + 48)
        nbrs[b, :lv + 1] = np.frombuffer(img, np.uint32, (lv + 1) * 2 * m, base + off + 48 + 4 * dim) \
            .reshape(lv + 1, 2 * m)
        dead[b] = flags == 3
    return dict(vecs=vecs, levels=levels, ncount=ncount, nbrs=nbrs, tids=tids, dead=dead, entry_point=entry,
                entry_level=entry_level, max_level=max_level, efc=efc, efs=efs, inserted=inserted)
