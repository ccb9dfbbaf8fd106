"""The reference's AM scan callbacks (include/ndb_am.h: ndb_ivfbeginscan / rescan / gettuple / endscan and the
hnsw four) driven the way PostgreSQL's executor drives ivf_am.c / hnsw_am.c: rescan with the ORDER BY key's
datum, gettuple until false.  Every tuple and distance must be the oracle's."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import ndbo
from tests.test_extract_vector import halfvec_datum, vector_datum
from tests.test_gpu_hnsw import build_graph, load
from tests.util import make_ivf_arrays, oracle_image

pytestmark = pytest.mark.gpu

VECTOR, HALFVEC = 0, 1


def _key(datum, strategy, kind=VECTOR):
    from neurondb_amd._lib import NdbScanKey
    buf = C.create_string_buffer(datum, len(datum)) if datum is not None else None
    k = NdbScanKey(strategy, kind, C.cast(buf, C.c_void_p) if buf is not None else None, len(datum) if datum else 0)
    return k, buf                                           # keep buf alive


def _drain(L, gettuple, scan):
    out = []
    while True:
        rc = gettuple(scan, 1)
        assert rc >= 0, L.ndbhip_last_error()
        if rc == 0:
            return out
        s = scan.contents
        out.append(((s.xs_heaptid.bi_hi, s.xs_heaptid.bi_lo, s.xs_heaptid.posid), s.xs_orderbyval, s.xs_orderbynull))


@pytest.fixture
def gucs():
    from neurondb_amd import _lib
    L = _lib.lib()
    yield lambda n, v: _lib.check(L.ndb_am_set_guc(n.encode(), v))
    for n, v in (("neurondb.ivf_probes", 10), ("neurondb.ivf_k", 10), ("neurondb.hnsw_ef_search", 64),
                 ("neurondb.hnsw_k", 10), ("neurondb.ref_compat", 0)):
        L.ndb_am_set_guc(n.encode(), v)


def test_ivf_scan_callbacks(gucs):
    from neurondb_amd import IvfIndex, _lib
    L = _lib.lib()
    a = make_ivf_arrays(5000, 64, 20, seed=61, dup_frac=0.1)
    img = oracle_image(a)
    ix = IvfIndex(64, 20)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    rng = np.random.default_rng(62)
    q = (a["rows"][17] + rng.standard_normal(64).astype(np.float32) * 0.05).astype(np.float32)
    scan = L.ndb_ivfbeginscan(ix._h, 0, 1)
    assert scan and L.ndb_ivfgettuple(scan, 1) == 0          # no rescan yet: no query, no tuple

    def run(strategy, datum, kind=VECTOR):
        key, keep = _key(datum, strategy, kind)
        _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(key), 1))
        return _drain(L, L.ndb_ivfgettuple, scan)

    def expect(query, strategy, nprobe, k, cap=0):
        et, ed, _ = img.search(query, strategy, nprobe, k, cap)
        return [((int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"])), d) for t, d in zip(et, ed)]

    for strategy in (1, 2, 3):                               # defaults: probes 10, k 10
        got = run(strategy, vector_datum(q))
        exp = expect(q, strategy, 10, 10)
        assert [g[0] for g in got] == [e[0] for e in exp]
        assert [np.float32(g[1]).view(np.uint32) for g in got] == [np.float32(e[1]).view(np.uint32) for e in exp]
        assert all(g[2] == 0 for g in got)
    gucs("neurondb.ivf_probes", 4)
    gucs("neurondb.ivf_k", 25)
    got = run(1, vector_datum(q))
    assert [g[0] for g in got] == [e[0] for e in expect(q, 1, 4, 25)] and len(got) == 25
    # a halfvec ORDER BY value goes through fp16_to_float like ivfExtractVectorData does
    h = (q * 0.5).astype(np.float16)
    lut_q = np.array([ndbo.lib().ndbo_fp16_to_float(int(v)) for v in h.view(np.uint16)], np.float32)
    got = run(3, halfvec_datum(h.view(np.uint16)), HALFVEC)
    assert [g[0] for g in got] == [e[0] for e in expect(lut_q, 3, 4, 25)]
    # reference-compatible mode: strategy 1 whatever the operator (Q1), nprobe 10, k 10 (Q3/Q4), k*10 cap (:1743)
    gucs("neurondb.ref_compat", 1)
    got = run(2, vector_datum(q))
    assert [g[0] for g in got] == [e[0] for e in expect(q, 1, 10, 10, cap=100)]
    _lib.check(L.ndbhip_ivf_set_nprobe(ix._h, 3))           # meta->nprobe / the reloption (:1487-1513)
    got = run(1, vector_datum(q))
    assert [g[0] for g in got] == [e[0] for e in expect(q, 1, 3, 10, cap=100)]
    _lib.check(L.ndbhip_ivf_set_nprobe(ix._h, 0))           # <= 0 on the page: the default
    got = run(1, vector_datum(q))
    assert [g[0] for g in got] == [e[0] for e in expect(q, 1, 10, 10, cap=100)]
    gucs("neurondb.ref_compat", 0)
    # wrong dimension: "does not match index dimension" -> no tuples (:1961-1972); NULL argument keeps the query
    assert run(1, vector_datum(q[:32])) == []
    _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(_key(vector_datum(q), 1)[0]), 1))
    nullkey, _ = _key(None, 1)
    _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(nullkey), 1))
    assert len(_drain(L, L.ndb_ivfgettuple, scan)) == 25
    # a malformed datum is the reference's ERROR: a negative code, not a crash
    bad = struct.pack("<ihh", 8 + 4 * 64, 70, 0) + q.tobytes()
    key, keep = _key(bad, 1)
    assert L.ndb_ivfrescan(scan, None, 0, C.byref(key), 1) < 0
    assert L.ndb_am_set_guc(b"neurondb.no_such", 1) < 0 and L.ndb_am_set_guc(b"neurondb.ivf_probes", 0) < 0
    L.ndb_ivfendscan(scan)


@pytest.mark.parametrize("compat", [1, 0], ids=["ref_compat", "intended"])
def test_hnsw_scan_callbacks(gucs, compat):
    """neurondb.ref_compat = 1: hnswgettuple's results are hnswSearch's (the reference's walk); 0 (the default): the
    `intended` search under the ORDER BY operator's strategy (oracle ndbo_h2_search_s) — same callbacks, same GUCs"""
    from neurondb_amd import _lib
    L = _lib.lib()
    g, vecs = build_graph(800, 32, 8, 40, seed=63)
    ix, a = load(g)
    gucs("neurondb.ref_compat", compat)
    rng = np.random.default_rng(64)
    q = rng.standard_normal(32).astype(np.float32)
    scan = L.ndb_hnswbeginscan(ix._h, 0, 1)
    assert scan and L.ndb_hnswgettuple(scan, 1) == 0

    def run(strategy, datum):
        key, keep = _key(datum, strategy)
        rc = L.ndb_hnswrescan(scan, None, 0, C.byref(key), 1)
        assert rc == 0, L.ndbhip_last_error()
        return key, keep

    def expect(query, strategy, ef, k):
        eb, ed, _ = g.search(query, strategy, ef, k) if compat else g.search_intended_s(query, strategy, ef, k)
        t = a["tids"][eb]
        return [tuple(int(x) for x in row) for row in np.asarray(t).reshape(len(eb), -1)]

    for strategy in (1, 2, 3):
        run(strategy, vector_datum(q))
        got = _drain(L, L.ndb_hnswgettuple, scan)
        assert [x[0] for x in got] == expect(q, strategy, 64, 10)
        assert all(x[2] == 1 for x in got)                   # hnswgettuple sets no ORDER BY value (Q13)
    gucs("neurondb.hnsw_ef_search", 16)
    gucs("neurondb.hnsw_k", 4)
    run(1, vector_datum(q))
    assert [x[0] for x in _drain(L, L.ndb_hnswgettuple, scan)] == expect(q, 1, 16, 4)
    # GUC <= 0: the meta page's efSearch decides (:923-936); the mirror carries it
    gucs("neurondb.hnsw_ef_search", 0)
    _lib.check(L.ndbhip_hnsw_set_meta(ix._h, 40, 24))
    efc, efs = C.c_int(), C.c_int()
    _lib.check(L.ndbhip_hnsw_get_meta(ix._h, C.byref(efc), C.byref(efs)))
    assert (efc.value, efs.value) == (40, 24)
    run(1, vector_datum(q))
    assert [x[0] for x in _drain(L, L.ndb_hnswgettuple, scan)] == expect(q, 1, 24, 4)
    gucs("neurondb.hnsw_ef_search", 16)
    # an operator strategy the AM does not know is hnswComputeDistance's ERROR (:1339-1343)
    run(4, vector_datum(q))
    assert L.ndb_hnswgettuple(scan, 1) < 0
    L.ndb_hnswendscan(scan)


def test_ivf_insert_and_bulkdelete_callbacks():
    """aminsert / ambulkdelete through the AM-level entry points: the entry lands in the list the oracle's
    insert-time rule picks, at its tail; VACUUM's callback sees every live heapPtr once and its hits vanish."""
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import BULKDELETE_CALLBACK, NdbItemPointer
    L = _lib.lib()
    a = make_ivf_arrays(3000, 32, 12, seed=71, dup_frac=0.05)
    ix = IvfIndex(32, 12)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    rng = np.random.default_rng(72)
    new = (a["rows"][rng.integers(0, 3000, 20)] + rng.standard_normal((20, 32)).astype(np.float32) * 0.1) \
        .astype(np.float32)
    exp_list = ndbo.ivf_assign_all(a["centroids"], new)
    for i in range(20):
        tid = NdbItemPointer(0, 900 + i, 1 + i)
        assert L.ndb_ivfinsert(ix._h, vector_datum(new[i]), 8 + 4 * 32, VECTOR, C.byref(tid)) == 1
    assert L.ndb_ivfinsert(ix._h, None, 0, VECTOR, C.byref(NdbItemPointer(0, 1, 1))) == 0      # NULL value
    assert L.ndb_ivfinsert(ix._h, vector_datum(new[0][:8]), 8 + 4 * 8, VECTOR, C.byref(NdbItemPointer(0, 1, 1))) < 0
    cent, ll, rows, tids = ix.export()
    assert np.array_equal(ll, np.asarray(a["list_len"]) + np.bincount(exp_list, minlength=12))
    off = np.concatenate([[0], np.cumsum(ll)])
    for lst in range(12):                                     # arrivals sit at the tail, in arrival order
        mine = [i for i in range(20) if exp_list[i] == lst]
        tail = tids[off[lst + 1] - len(mine):off[lst + 1]]
        assert [int(t["bi_lo"]) for t in tail] == [900 + i for i in mine]
    seen = []

    def cb(ip, state):
        t = ip.contents
        seen.append((t.bi_hi, t.bi_lo, t.posid))
        return 1 if (t.bi_lo >= 900 or t.posid % 7 == 0) else 0
    removed = C.c_int64(0)
    _lib.check(L.ndb_ivfbulkdelete(ix._h, BULKDELETE_CALLBACK(cb), None, C.byref(removed)))
    assert len(seen) == 3020 and len(set(seen)) == len(set((int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"]))
                                                          for t in tids))
    kept = [t for t in tids if not (t["bi_lo"] >= 900 or t["posid"] % 7 == 0)]
    assert removed.value == 3020 - len(kept)
    assert [tuple(t) for t in ix.export()[3].tolist()] == [tuple(t) for t in np.array(kept).tolist()]


@pytest.mark.parametrize("compat", [1, 0], ids=["ref_compat", "intended"])
def test_hnsw_insert_and_bulkdelete_callbacks(gucs, compat):
    """neurondb.ref_compat = 1: ndb_hnswinsert is hnswInsertNode (oracle ndbo_hnsw_insert); 0: the `intended` insert — one row,
    one batch of the definition's schedule (oracle ndbo_h2_build on the graph as it stands) — on top of a graph the
    reference's own insert built: the mirror equals the model slot for slot either way"""
    from neurondb_amd import HnswIndex, _lib
    from neurondb_amd._lib import BULKDELETE_CALLBACK, NdbItemPointer
    L = _lib.lib()
    gucs("neurondb.ref_compat", compat)
    g, vecs = build_graph(400, 16, 5, 200, seed=73)           # ef_construction 200: what ndb_hnswinsert uses
    ix, a = load(g)
    # model = the oracle over what the loaded (packed) mirror holds
    e = ix.export()
    model = ndbo.HnswGraph.from_arrays(a["vecs"], e["levels"], e["ncount"], e["nbrs"], a["tids"], e["entry_point"],
                                       e["entry_level"], 5, 200, cap_nodes=500)
    rng = np.random.default_rng(74)
    more = rng.standard_normal((30, 16)).astype(np.float32)
    for i in range(30):
        lv = L.ndb_hnsw_level_from_uniform(float(rng.uniform(1e-9, 1.0)), np.float32(0.36))
        t = ndbo.tids_from_rows(np.array([400 + i]))[0]
        tid = NdbItemPointer(int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"]))
        assert L.ndb_hnswinsert(ix._h, vector_datum(more[i]), 8 + 4 * 16, VECTOR, C.byref(tid), lv) == 1
        if compat:
            model.insert(more[i], 400 + i, lv)
        else:
            assert model.build_intended(more[i:i + 1], [lv], tids=ndbo.tids_from_rows(np.array([400 + i])), batch_div=64,
                                        batch_max=1024, select=1) == 1
    q = rng.standard_normal((8, 16)).astype(np.float32)
    from tests.test_gpu_hnsw import check
    ix.nblocks = 431
    d, e2 = ix.export(), model.arrays()
    assert d["nblocks"] == 431 and (d["entry_point"], d["entry_level"]) == (e2["entry_point"], e2["entry_level"])
    assert np.array_equal(d["levels"][1:], e2["levels"][1:])
    if not compat:
        # (the reference's inserts write lists above a node's own level, Q12 / Q21 — kept by the dense layouts, not part of the
        # index: compared up to every node's level)
        for b_ in range(1, 431):
            top = int(d["levels"][b_]) + 1
            assert np.array_equal(d["ncount"][b_, :top], e2["ncount"][b_, :top]), b_
            assert np.array_equal(d["nbrs"][b_, :top], e2["nbrs"][b_, :top]), b_
    for strategy in (1, 2):
        check(model, ix, q, strategy, 32, 10)
    # ... and the scan the same GUC selects finds the same rows as its oracle
    for strategy in (1, 2, 3):
        ob = ix.search_intended(q, 32, 10, strategy=strategy)
        for i in range(len(q)):
            eb, ed, ns = model.search_intended_s(q[i], strategy, 32, 10)
            assert np.array_equal(ob[0][i, :ob[2][i]], eb) and np.array_equal(ob[1][i, :ob[2][i]].view(np.uint32), ed.view(np.uint32))

    def cb(ip, state):
        return 1 if ip.contents.posid % 9 == 0 else 0
    removed = C.c_int64(0)
    _lib.check(L.ndb_hnswbulkdelete(ix._h, BULKDELETE_CALLBACK(cb), None, C.byref(removed)))
    b = model.arrays()
    victims = np.array([t for t in ndbo.tids_from_rows(np.arange(430)) if t["posid"] % 9 == 0])
    assert model.bulkdelete(victims) == removed.value > 0
    for strategy in (1, 3):
        check(model, ix, q, strategy, 32, 10)
    _lib.check(L.ndb_hnswbulkdelete(ix._h, BULKDELETE_CALLBACK(cb), None, C.byref(removed)))
    assert removed.value == 0                                 # dead line pointers are not offered again


def test_plain_c_program_drives_the_abi():
    """examples/c_demo.c — ambuild, the AM scan callbacks, aminsert and ambulkdelete for both AMs from a C
    program that links libndbhip.so and nothing else (no Python, no torch in that process)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "neurondb_amd", "lib", "c_demo")
    assert os.path.exists(exe), "build it: make -C neurondb_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c_demo: OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
