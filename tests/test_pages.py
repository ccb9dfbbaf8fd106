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
