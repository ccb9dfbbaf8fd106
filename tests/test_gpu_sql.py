"""The SQL-level batch / "GPU" entry points (include/ndb_sql.h) called the way their PG_FUNCTION
wrappers would call them: vector datums in, real[] / rows out.  Distances must be the oracle's restatement of
src/vector/vector_distance.c bit for bit; kNN rows must be the access method's (oracle search)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ndbo
from tests.test_extract_vector import vector_datum
from tests.test_gpu_hnsw import build_graph, load
from tests.util import make_ivf_arrays, oracle_image

pytestmark = pytest.mark.gpu


def _datum_array(datums):
    """(pointer array, length array, keep-alive) for a vector[] argument; None = a NULL element"""
    bufs = [None if d is None else C.create_string_buffer(d, len(d)) for d in datums]
    ptrs = (C.c_void_p * len(datums))(*[None if b is None else C.cast(b, C.c_void_p) for b in bufs])
    lens = (C.c_size_t * len(datums))(*[0 if d is None else len(d) for d in datums])
    return ptrs, lens, bufs


def test_distance_batch_functions_match_the_operator_kernels():
    from neurondb_amd import _lib
    _lib.ensure_init()
    L, O = _lib.lib(), ndbo.lib()
    rng = np.random.default_rng(7)
    fns = {1: (L.ndb_vector_l2_distance_batch, O.ndbo_op_l2), 2: (L.ndb_vector_cosine_distance_batch, O.ndbo_op_cosine),
           3: (L.ndb_vector_inner_product_batch, O.ndbo_op_ip)}
    for dim in (3, 16, 100, 768):
        q = rng.standard_normal(dim).astype(np.float32)
        vecs = [(rng.standard_normal(dim) * s).astype(np.float32) for s in rng.choice([1e-3, 1.0, 1e3], 70)]
        vecs[4] = np.zeros(dim, np.float32)                     # cosine: zero norm -> 1.0
        vecs[9] = q.copy()
        datums = [vector_datum(v) for v in vecs]
        datums[11] = None                                       # NULL element -> 0.0 (construct_array ignores nulls[])
        datums[20] = vector_datum(np.ones(dim + 1, np.float32))  # other dimension -> 0.0
        ptrs, lens, keep = _datum_array(datums)
        qd = vector_datum(q)
        for s, (fn, ofn) in fns.items():
            out = np.full(len(datums), 7.0, np.float32)
            _lib.check(fn(ptrs, lens, len(datums), qd, len(qd), out.ctypes.data))
            exp = np.array([0.0 if i in (11, 20) else ofn(vecs[i], q, dim, 0) for i in range(len(vecs))], np.float32)
            assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (dim, s)
    # the reference's ERRORs
    out = np.zeros(4, np.float32)
    ptrs, lens, keep = _datum_array([vector_datum(np.ones(4, np.float32))])
    qd = vector_datum(np.ones(4, np.float32))
    assert L.ndb_vector_l2_distance_batch(ptrs, lens, 0, qd, len(qd), out.ctypes.data) < 0      # empty array
    assert b"must not be empty" in L.ndbhip_last_error()
    assert L.ndb_vector_l2_distance_batch(ptrs, lens, 1, None, 0, out.ctypes.data) < 0          # NULL query
    big = np.full(4, 3e38, np.float32)
    ptrs, lens, keep = _datum_array([vector_datum(-big)])
    qd = vector_datum(big)
    assert L.ndb_vector_l2_distance_batch(ptrs, lens, 1, qd, len(qd), out.ctypes.data) < 0      # (float4) sqrt = inf
    assert b"NaN or Infinity" in L.ndbhip_last_error()
    _lib.check(L.ndb_vector_inner_product_batch(ptrs, lens, 1, qd, len(qd), out.ctypes.data))   # IP does not check
    assert np.isinf(out[0])
    zero = vector_datum(np.zeros(0, np.float32))
    ptrs, lens, keep = _datum_array([zero])
    qd = vector_datum(np.ones(4, np.float32))
    assert L.ndb_vector_l2_distance_batch(ptrs, lens, 1, qd, len(qd), out.ctypes.data) < 0      # dim 0 element
    assert b"invalid vector dimension" in L.ndbhip_last_error()


def test_distance_gpu_functions_return_what_the_cpu_functions_return():
    from neurondb_amd import _lib
    _lib.ensure_init()
    L, O = _lib.lib(), ndbo.lib()
    rng = np.random.default_rng(8)
    for dim in (5, 128, 1536):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        da, db = vector_datum(a), vector_datum(b)
        r = C.c_float()
        _lib.check(L.ndb_vector_l2_distance_gpu(da, len(da), db, len(db), C.byref(r)))
        assert np.float32(r.value).view(np.uint32) == np.float32(O.ndbo_op_l2(a, b, dim, 0)).view(np.uint32)
        _lib.check(L.ndb_vector_cosine_distance_gpu(da, len(da), db, len(db), C.byref(r)))
        assert np.float32(r.value).view(np.uint32) == np.float32(O.ndbo_op_cosine(a, b, dim, 0)).view(np.uint32)
        _lib.check(L.ndb_vector_inner_product_gpu(da, len(da), db, len(db), C.byref(r)))
        # inner_product_distance = (float4) (-sum)   (vector_distance.c:145-157); ndbo_op_ip is the operator's +dot
        assert np.float32(r.value).view(np.uint32) == np.float32(-O.ndbo_op_ip(a, b, dim, 0)).view(np.uint32)
    short = vector_datum(np.ones(3, np.float32))
    assert L.ndb_vector_l2_distance_gpu(da, len(da), short, len(short), C.byref(r)) < 0
    assert b"dimensions must match" in L.ndbhip_last_error()


def _rows(buf, n):
    return [(r.query_no, (r.heaptid.bi_hi, r.heaptid.bi_lo, r.heaptid.posid), r.id, np.float32(r.distance))
            for r in buf[:n]]


def test_ivf_knn_search_gpu_rows_equal_the_index_scan():
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import NdbKnnRow
    L = _lib.lib()
    a = make_ivf_arrays(3000, 64, 20, seed=71, dup_frac=0.1)
    img = oracle_image(a)
    ix = IvfIndex(64, 20)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    rng = np.random.default_rng(72)
    qs = [a["rows"][rng.integers(0, 3000)] + rng.standard_normal(64).astype(np.float32) * 0.1 for _ in range(9)]
    datums = [vector_datum(q) for q in qs]
    datums[3] = None                                        # NULL element: no rows
    datums[6] = vector_datum(np.ones(32, np.float32))       # other dimension: every entry is skipped, no rows
    ptrs, lens, keep = _datum_array(datums)
    for strategy, k, nprobe in ((1, 10, 5), (2, 7, 20), (3, 10, 1000)):
        buf = (NdbKnnRow * (len(datums) * k))()
        n = C.c_int64()
        _lib.check(L.ndb_ivf_knn_search_gpu(ix._h, strategy, ptrs, lens, len(datums), k, nprobe, buf, C.byref(n)))
        exp = []
        for i, q in enumerate(qs):
            if i in (3, 6):
                continue
            et, ed, _ = img.search(q, strategy, nprobe, k, 0)
            exp += [(i, (int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"])), (int(t["bi_hi"]) << 16) | int(t["bi_lo"]),
                     np.float32(d)) for t, d in zip(et, ed)]
        got = _rows(buf, n.value)
        assert [g[:3] for g in got] == [e[:3] for e in exp]
        assert np.array_equal(np.array([g[3] for g in got]).view(np.uint32), np.array([e[3] for e in exp]).view(np.uint32))
    buf = (NdbKnnRow * 10)()
    n = C.c_int64()
    assert L.ndb_ivf_knn_search_gpu(ix._h, 1, ptrs, lens, len(datums), 0, 5, buf, C.byref(n)) < 0
    assert b"k must be between" in L.ndbhip_last_error()
    assert L.ndb_ivf_knn_search_gpu(ix._h, 1, ptrs, lens, len(datums), 1, 1001, buf, C.byref(n)) < 0
    assert b"nprobe must be between 1 and 1000" in L.ndbhip_last_error()
    assert L.ndb_ivf_knn_search_gpu(None, 1, ptrs, lens, len(datums), 1, 5, buf, C.byref(n)) < 0


@pytest.mark.parametrize("compat", [1, 0], ids=["ref_compat", "intended"])
def test_hnsw_knn_search_gpu_rows_equal_the_index_scan(compat):
    from neurondb_amd import _lib
    from neurondb_amd._lib import NdbKnnRow
    L = _lib.lib()
    g, vecs = build_graph(900, 32, 8, 40, seed=73)
    ix, a = load(g)
    _lib.check(L.ndb_am_set_guc(b"neurondb.ref_compat", compat))
    try:
        _hnsw_knn_rows(L, _lib, NdbKnnRow, g, ix, a, compat)
    finally:
        _lib.check(L.ndb_am_set_guc(b"neurondb.ref_compat", 0))


def _hnsw_knn_rows(L, _lib, NdbKnnRow, g, ix, a, compat):
    rng = np.random.default_rng(74)
    qs = [rng.standard_normal(32).astype(np.float32) for _ in range(6)]
    datums = [vector_datum(q) for q in qs]
    datums[2] = None
    ptrs, lens, keep = _datum_array(datums)
    for strategy, k, ef in ((1, 10, 100), (2, 5, 64), (3, 10, 16)):
        buf = (NdbKnnRow * (len(datums) * k))()
        n = C.c_int64()
        _lib.check(L.ndb_hnsw_knn_search_gpu(ix._h, strategy, ptrs, lens, len(datums), k, ef, buf, C.byref(n)))
        exp = []
        for i, q in enumerate(qs):
            if i == 2:
                continue
            eb, ed, _ = g.search(q, strategy, ef, k) if compat else g.search_intended_s(q, strategy, ef, k)
            t = np.asarray(a["tids"][eb])
            exp += [(i, tuple(int(x) for x in np.asarray(row).reshape(-1)), np.float32(d))
                    for row, d in zip(t.reshape(len(eb), -1), ed)]
        got = _rows(buf, n.value)
        assert [(g_[0], g_[1]) for g_ in got] == [(e[0], e[1]) for e in exp]
        assert np.array_equal(np.array([g_[3] for g_ in got]).view(np.uint32),
                              np.array([e[2] for e in exp]).view(np.uint32))
    buf = (NdbKnnRow * 10)()
    n = C.c_int64()
    assert L.ndb_hnsw_knn_search_gpu(ix._h, 1, ptrs, lens, len(datums), 10, 0, buf, C.byref(n)) < 0
    assert b"ef_search must be between" in L.ndbhip_last_error()


def test_ivf_knn_search_gpu_with_a_large_query_array_takes_the_screened_scan():
    """150 queries in one call: the batch is large enough for the screened scan (bound pass + the reference's
    arithmetic for the survivors); rows still equal the oracle's index scan, for all three operator classes."""
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import NdbKnnRow
    L = _lib.lib()
    a = make_ivf_arrays(5000, 64, 16, seed=81, dup_frac=0.1)
    img = oracle_image(a)
    ix = IvfIndex(64, 16)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    rng = np.random.default_rng(82)
    qs = [a["rows"][rng.integers(0, 5000)] + rng.standard_normal(64).astype(np.float32) * 0.1 for _ in range(150)]
    ptrs, lens, keep = _datum_array([vector_datum(q) for q in qs])
    k, nprobe = 10, 6
    for strategy in (1, 2, 3):
        _lib.check(L.ndbhip_stats_reset())
        buf = (NdbKnnRow * (len(qs) * k))()
        n = C.c_int64()
        _lib.check(L.ndb_ivf_knn_search_gpu(ix._h, strategy, ptrs, lens, len(qs), k, nprobe, buf, C.byref(n)))
        assert _lib.stats()["rows_rescored"] > 0                      # the second pass ran
        got = _rows(buf, n.value)
        exp = []
        for i, q in enumerate(qs):
            et, ed, _ = img.search(q, strategy, nprobe, k, 0)
            exp += [(i, (int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"])), np.float32(d)) for t, d in zip(et, ed)]
        assert [(g[0], g[1]) for g in got] == [(e[0], e[1]) for e in exp]
        assert np.array_equal(np.array([g[3] for g in got]).view(np.uint32), np.array([e[2] for e in exp]).view(np.uint32))
