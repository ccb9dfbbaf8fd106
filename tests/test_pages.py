"""Page codec (index pages <-> flat mirror arrays) against an independent writer/reader of the
reference's on-disk format.  CPU only (the codec is host code)."""
import ctypes as C

import numpy as np
import pytest

from neurondb_amd import _lib
from tests import pgpages


def unpack_with_product(img):
    L = _lib.lib()
    a = np.frombuffer(img, np.uint8).copy()
    nb = len(img) // 8192
    dim, nl, nc, ver = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    live = C.c_int64()
    _lib.check(L.ndbhip_ivf_pages_info(a.ctypes.data, nb, C.byref(dim), C.byref(nl), C.byref(nc), C.byref(live),
                                       C.byref(ver)))
    cent = np.zeros((nc.value, dim.value), np.float32)
    ll = np.zeros(nc.value, np.int64)
    rows = np.zeros((max(live.value, 1), dim.value), np.float32)
    t6 = np.zeros((max(live.value, 1), 6), np.uint8)
    _lib.check(L.ndbhip_ivf_pages_unpack(a.ctypes.data, nb, cent.ctypes.data, ll.ctypes.data, rows.ctypes.data,
                                         t6.ctypes.data))
    return dim.value, cent, ll, rows[:live.value], t6[:live.value], ver.value, nl.value


def make_lists(rng, nl, dim, sizes):
    lists = []
    r = 0
    for L in range(nl):
        ent = []
        for _ in range(sizes[L]):
            tid = np.array([r >> 16, r & 0xFFFF, r % 200 + 1], np.uint16).view(np.uint8)
            ent.append((rng.standard_normal(dim).astype(np.float32), tid.tobytes()))
            r += 1
        lists.append(ent)
    return lists


@pytest.mark.parametrize("dim,nl,sizes", [(4, 10, [0, 5, 300, 1, 17, 0, 64, 2, 999, 3]), (28, 5, [40, 0, 400, 7, 90]),
                                           (128, 7, [31, 2, 0, 100, 16, 15, 1])])
def test_reads_reference_format_pages_with_dead_and_foreign_entries(dim, nl, sizes):
    rng = np.random.default_rng(dim)
    cents = rng.standard_normal((nl, dim)).astype(np.float32)
    lists = make_lists(rng, nl, dim, sizes)
    big = int(np.argmax(sizes))
    dead = {(big, 0), (big, 7), (big, sizes[big] - 1)}
    foreign = {(big, 3)}
    img = pgpages.write_reference_format(cents, lists, dead=dead, foreign_dim=foreign)
    d, c, ll, rows, t6, ver, nlm = unpack_with_product(img)
    assert (d, ver, nlm) == (dim, 1, nl)
    assert np.array_equal(c.view(np.uint32), cents.view(np.uint32))
    exp_len = list(sizes)
    exp_len[big] -= len(dead) + len(foreign)
    assert list(ll) == exp_len
    # rows in chain order, skipping dead / foreign-dim entries
    exp_rows, exp_t = [], []
    for L in range(nl):
        for i, (v, t) in enumerate(lists[L]):
            if (L, i) in dead or (L, i) in foreign:
                continue
            exp_rows.append(v)
            exp_t.append(np.frombuffer(t, np.uint8))
    assert np.array_equal(rows.view(np.uint32), np.array(exp_rows, np.float32).reshape(-1, dim).view(np.uint32))
    assert np.array_equal(t6, np.array(exp_t, np.uint8).reshape(-1, 6))
    # and the independent reader agrees with itself
    d2, c2, ll2, rows2, t62, _ = pgpages.read_image(img)
    assert np.array_equal(rows2, rows) and np.array_equal(ll2, ll)


@pytest.mark.parametrize("dim,nl", [(128, 15), (128, 100), (768, 40), (1536, 3), (4, 185), (4, 186)])
def test_pack_then_unpack_roundtrip_and_version_rule(dim, nl):
    """writer: version 1 while the centroids fit one page (bit-compatible with the reference:
    185 @ dim 4, 15 @ dim 128, 2 @ dim 768 — quirk Q6), version 2 (chained centroid pages) beyond."""
    rng = np.random.default_rng(nl)
    cents = rng.standard_normal((nl, dim)).astype(np.float32)
    ll = rng.integers(0, 40, nl).astype(np.int64)
    ll[0] = 0
    n = int(ll.sum())
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    t6 = rng.integers(0, 256, (n, 6)).astype(np.uint8)
    L = _lib.lib()
    need = L.ndbhip_ivf_pages_needed(dim, nl, ll.ctypes.data)
    assert need > 0
    pages = np.zeros(need * 8192, np.uint8)
    nb = C.c_uint32()
    _lib.check(L.ndbhip_ivf_pages_pack(dim, nl, 10, nl, cents.ctypes.data, ll.ctypes.data, rows.ctypes.data,
                                       t6.ctypes.data, pages.ctypes.data, need, C.byref(nb)))
    img = pages[: nb.value * 8192].tobytes()
    csize = (24 + 4 * dim + 7) & ~7
    fits_one = nl * (csize + 4) <= 8192 - 24 - 24
    d, c, ll2, rows2, t62, ver, _ = unpack_with_product(img)
    assert ver == (1 if fits_one else 2)
    assert np.array_equal(c.view(np.uint32), cents.view(np.uint32)) and np.array_equal(ll2, ll)
    assert np.array_equal(rows2.view(np.uint32), rows.view(np.uint32)) and np.array_equal(t62, t6)
    # the independent reader decodes the product's pages identically
    d3, c3, ll3, rows3, t63, ver3 = pgpages.read_image(img)
    assert np.array_equal(c3.view(np.uint32), cents.view(np.uint32)) and np.array_equal(ll3, ll)
    assert np.array_equal(rows3.view(np.uint32), rows.view(np.uint32)) and np.array_equal(t63, t6)


def test_bad_images_are_errors():
    L = _lib.lib()
    z = np.zeros(8192, np.uint8)
    assert L.ndbhip_ivf_pages_info(z.ctypes.data, 1, None, None, None, None, None) == _lib.ERR_INVALID
    img = bytearray(pgpages.write_reference_format(np.zeros((2, 4), np.float32), [[], []]))
    img[24:28] = b"\0\0\0\0"          # wrong magic
    a = np.frombuffer(bytes(img), np.uint8).copy()
    assert L.ndbhip_ivf_pages_info(a.ctypes.data, len(a) // 8192, None, None, None, None, None) == _lib.ERR_INVALID
    assert "magic" in _lib.last_error()


# ---------------------------------------------------------------------------------------------
# hnsw relation pages
# ---------------------------------------------------------------------------------------------

def _small_graph(n=120, dim=12, m=5, efc=16, seed=4):
    from oracle import ndbo
    rng = np.random.default_rng(seed)
    L = ndbo.lib()
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    for i in range(n):
        lv = 3 if i == 9 else L.ndbo_hnsw_level_from_uniform(float(rng.uniform(1e-9, 1)), np.float32(0.36))
        g.insert(vecs[i], i, int(lv))
    a = g.arrays()
    # what a page can hold: slots of levels 0..level (the Q12/Q21 out-of-node writes are not on the item)
    for b in range(1, a["nblocks"]):
        a["nbrs"][b, a["levels"][b] + 1:] = 0xFFFFFFFF
    a["tids6"] = np.ascontiguousarray(a["tids"]).view(np.uint8).reshape(-1, 6)
    return g, a


def test_hnsw_pages_unpack_reads_the_reference_layout():
    import ctypes as C
    from neurondb_amd import _lib
    g, a = _small_graph()
    nb, m, dim = a["nblocks"], a["m"], a["dim"]
    dead = {7, 30}
    img = pgpages.write_hnsw_reference_format(a["vecs"], a["levels"], a["ncount"], a["nbrs"], a["tids6"],
                                              a["entry_point"], a["entry_level"], m, efc=16, efs=40, dead=dead)
    L = _lib.lib()
    buf = np.frombuffer(img, np.uint8)
    d, mm, efc, efs, ep, el = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_uint32(), C.c_int()
    _lib.check(L.ndbhip_hnsw_pages_info(buf.ctypes.data, nb, C.byref(d), C.byref(mm), C.byref(efc), C.byref(efs),
                                        C.byref(ep), C.byref(el)))
    assert (d.value, mm.value, efc.value, efs.value, ep.value, el.value) == \
        (dim, m, 16, 40, a["entry_point"], a["entry_level"])
    vecs = np.zeros((nb, dim), np.float32)
    levels = np.zeros(nb, np.int32)
    ncount = np.zeros((nb, 16), np.int16)
    nbrs = np.zeros((nb, 16, 2 * m), np.uint32)
    tids = np.zeros((nb, 6), np.uint8)
    dd = np.zeros(nb, np.uint8)
    _lib.check(L.ndbhip_hnsw_pages_unpack(buf.ctypes.data, nb, vecs.ctypes.data, levels.ctypes.data,
                                          ncount.ctypes.data, nbrs.ctypes.data, tids.ctypes.data, dd.ctypes.data))
    assert np.array_equal(vecs[1:], a["vecs"][1:]) and np.array_equal(levels[1:], a["levels"][1:])
    assert np.array_equal(ncount[1:], a["ncount"][1:]) and np.array_equal(nbrs[1:], a["nbrs"][1:])
    assert np.array_equal(tids[1:], a["tids6"][1:])
    assert set(np.nonzero(dd)[0].tolist()) == dead
    # a truncated item or a foreign magic number is refused, not read
    bad = bytearray(img)
    bad[24:28] = b"\0\0\0\0"
    assert L.ndbhip_hnsw_pages_info(np.frombuffer(bytes(bad), np.uint8).ctypes.data, nb, None, None, None, None,
                                    None, None) != 0


def test_hnsw_pages_pack_is_readable_by_an_independent_reader():
    from neurondb_amd import _lib
    g, a = _small_graph(seed=6)
    nb, m, dim = a["nblocks"], a["m"], a["dim"]
    dead = np.zeros(nb, np.uint8)
    dead[[3, 44]] = 1
    pages = np.zeros(nb * pgpages.BLCKSZ, np.uint8)
    _lib.check(_lib.lib().ndbhip_hnsw_pages_pack(
        dim, m, 16, 40, nb, a["vecs"].ctypes.data, a["levels"].ctypes.data, a["ncount"].ctypes.data,
        a["nbrs"].ctypes.data, a["tids6"].ctypes.data, dead.ctypes.data, a["entry_point"], a["entry_level"],
        pages.ctypes.data, nb))
    r = pgpages.read_hnsw_image(pages.tobytes(), m)
    assert np.array_equal(r["vecs"][1:], a["vecs"][1:]) and np.array_equal(r["levels"][1:], a["levels"][1:])
    assert np.array_equal(r["ncount"][1:], a["ncount"][1:]) and np.array_equal(r["nbrs"][1:], a["nbrs"][1:])
    assert np.array_equal(r["tids"][1:], a["tids6"][1:]) and np.array_equal(r["dead"], dead)
    assert (r["entry_point"], r["entry_level"], r["efc"], r["efs"]) == (a["entry_point"], a["entry_level"], 16, 40)
    assert r["max_level"] == int(a["levels"][1:].max()) and r["inserted"] == nb - 1 - 2
    # byte-identical to the image the independent writer produces for the same graph
    img = pgpages.write_hnsw_reference_format(a["vecs"], a["levels"], a["ncount"], a["nbrs"], a["tids6"],
                                              a["entry_point"], a["entry_level"], m, efc=16, efs=40, dead={3, 44})
    assert pages.tobytes() == img
