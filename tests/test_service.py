"""The device-owner service (include/ndb_service.h, csrc/ndb_service.cpp): many backend processes hand single
queries to a shared-memory ring, one owner coalesces them into batches.

CPU part (runs without a GPU): 16 backend processes against an owner loop driven from Python through
ndb_service_poll / ndb_service_complete with the ORACLE as the executor (test infrastructure standing in for the
device) — the ring, the slot life cycle, batching by parameter set, wake-ups and result routing are the product
code under test.  GPU part: the real executor (ndb_service_serve_ivf) in its own process, results == the oracle,
and the aggregate rate of concurrent backends far above what one backend gets on its own."""
import ctypes as C
import multiprocessing as mp
import os
import threading
import time

import numpy as np
import pytest

from tests.util import make_ivf_arrays, oracle_image


def _backend(rank, name, dim, nq, k, ret):
    from neurondb_amd import _lib
    L = _lib.lib()
    c = C.c_void_p()
    for _ in range(200):
        if L.ndb_client_connect(name.encode(), C.byref(c)) == 0:
            break
        time.sleep(0.02)
    rng = np.random.default_rng(100 + rank)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    out = []
    for i in range(nq):
        tids = np.zeros((k, 6), np.uint8)
        dist = np.zeros(k, np.float32)
        n = C.c_int()
        nprobe = 3 + (rank + i) % 2                       # two parameter sets in flight at once
        rc = L.ndb_client_search(c, q[i].ctypes.data_as(C.c_void_p), 1, nprobe, k, 0, tids.ctypes.data_as(C.c_void_p),
                                 dist.ctypes.data_as(C.c_void_p), C.byref(n), 20000)
        out.append((rc, nprobe, n.value, tids.copy(), dist.copy()))
    L.ndb_client_disconnect(c)
    ret[rank] = (q, out)


def test_sixteen_backends_one_owner_results_routed_and_batched():
    from neurondb_amd import _lib
    from oracle import ndbo
    L = _lib.lib()
    dim, k, nback, nq = 24, 5, 16, 12
    a = make_ivf_arrays(1500, dim, 8, seed=9)
    img = oracle_image(a)
    name = f"/ndb_service_test_{os.getpid()}"
    s = C.c_void_p()
    _lib.check(L.ndb_service_create(name.encode(), dim, 16, 64, C.byref(s)))
    batches = []

    def owner():
        ids = (C.c_int * 64)()
        qbuf = np.zeros((64, dim), np.float32)
        st, npb, kk, cap = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        while not L.ndb_service_stopped(s):
            n = L.ndb_service_poll(s, 64, 20000, 2000, ids, qbuf.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(npb),
                                   C.byref(kk), C.byref(cap))
            if n <= 0:
                continue
            batches.append((n, npb.value))
            tids = np.zeros((n, kk.value, 6), np.uint8)
            dist = np.zeros((n, kk.value), np.float32)
            cnt = np.zeros(n, np.int32)
            for i in range(n):                            # the executor of this test: the CPU oracle
                t, d, _ = img.search(qbuf[i], st.value, npb.value, kk.value, cap.value)
                cnt[i] = len(t)
                tids[i, :len(t)] = np.frombuffer(t.tobytes(), np.uint8).reshape(-1, 6)
                dist[i, :len(t)] = d
            _lib.check(L.ndb_service_complete(s, n, ids, tids.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
                                              cnt.ctypes.data_as(C.c_void_p), kk.value, 0))

    th = threading.Thread(target=owner)
    th.start()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_backend, args=(r, name, dim, nq, k, ret)) for r in range(nback)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    L.ndb_service_stop(s)
    th.join(30)
    L.ndb_service_destroy(s)
    assert len(ret) == nback
    for r in range(nback):
        q, out = ret[r]
        for i, (rc, nprobe, n, tids, dist) in enumerate(out):
            assert rc == 0
            et, ed, _ = img.search(q[i], 1, nprobe, k, 0)
            assert n == len(et)
            assert np.array_equal(ndbo.tids_to_u64(tids[:n].copy().view(ndbo.TID_DTYPE).reshape(-1)), ndbo.tids_to_u64(et))
            assert np.array_equal(dist[:n].view(np.uint32), ed.view(np.uint32))
    assert sum(n for n, _ in batches) == nback * nq
    assert max(n for n, _ in batches) > 1                 # concurrent backends really were coalesced
    assert {p for _, p in batches} == {3, 4}              # and never across parameter sets


def test_index_am_callbacks_answer_through_the_service():
    """ndb_ivfbeginscan(NULL) / rescan / gettuple in a process that never touches a device or a mirror: with
    neurondb.device_service set the first gettuple goes through the ring (csrc/ndb_am.cpp)."""
    from neurondb_amd import _lib
    from oracle import ndbo
    L = _lib.lib()
    dim, k = 16, 10
    a = make_ivf_arrays(800, dim, 6, seed=19)
    img = oracle_image(a)
    name = f"/ndb_service_am_{os.getpid()}"
    s = C.c_void_p()
    _lib.check(L.ndb_service_create(name.encode(), dim, 16, 8, C.byref(s)))

    def owner():
        ids = (C.c_int * 8)()
        qbuf = np.zeros((8, dim), np.float32)
        st, npb, kk, cap = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        while not L.ndb_service_stopped(s):
            n = L.ndb_service_poll(s, 8, 20000, 100, ids, qbuf.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(npb),
                                   C.byref(kk), C.byref(cap))
            for i in range(max(n, 0)):
                t, d, _ = img.search(qbuf[i], st.value, npb.value, kk.value, cap.value)
                tids = np.zeros((1, kk.value, 6), np.uint8)
                dist = np.zeros((1, kk.value), np.float32)
                tids[0, :len(t)] = np.frombuffer(t.tobytes(), np.uint8).reshape(-1, 6)
                dist[0, :len(t)] = d
                cnt = np.array([len(t)], np.int32)
                one = (C.c_int * 1)(ids[i])
                _lib.check(L.ndb_service_complete(s, 1, one, tids.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
                                                  cnt.ctypes.data_as(C.c_void_p), kk.value, 0))

    th = threading.Thread(target=owner)
    th.start()
    try:
        _lib.check(L.ndb_am_use_service(name.encode()))
        _lib.check(L.ndb_am_set_guc(b"neurondb.ivf_probes", 4))
        scan = L.ndb_ivfbeginscan(None, 0, 1)
        assert scan
        q = np.random.default_rng(3).standard_normal(dim).astype(np.float32)
        from tests.test_extract_vector import vector_datum
        from tests.test_gpu_am import VECTOR
        datum = vector_datum(q)
        buf = C.create_string_buffer(datum, len(datum))
        key = _lib.NdbScanKey(1, VECTOR, C.cast(buf, C.c_void_p), len(datum))
        _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(key), 1))
        got_t, got_d = [], []
        while L.ndb_ivfgettuple(scan, 1) == 1:
            tp = scan.contents.xs_heaptid
            got_t.append((tp.bi_hi, tp.bi_lo, tp.posid))
            got_d.append(scan.contents.xs_orderbyval)
        L.ndb_ivfendscan(scan)
        et, ed, _ = img.search(q, 1, 4, 10, 0)
        assert got_t == [(int(t["bi_hi"]), int(t["bi_lo"]), int(t["posid"])) for t in et]
        assert np.array_equal(np.array(got_d, np.float32).view(np.uint32), ed.view(np.uint32))
    finally:
        L.ndb_am_use_service(None)
        L.ndb_service_stop(s)
        th.join(30)
        L.ndb_service_destroy(s)
        _lib.check(L.ndb_am_set_guc(b"neurondb.ivf_probes", 10))


def test_a_backend_without_a_service_gets_nodevice_and_can_fall_back():
    from neurondb_amd import _lib
    L = _lib.lib()
    c = C.c_void_p()
    rc = L.ndb_client_connect(b"/ndb_service_that_does_not_exist", C.byref(c))
    assert rc == _lib.NDBHIP_ERR_NODEVICE if hasattr(_lib, "NDBHIP_ERR_NODEVICE") else rc < 0


@pytest.mark.gpu
def test_device_owner_serves_concurrent_backends_faster_than_one_backend_alone():
    from tools.service_bench import run
    one = run(backends=1, inflight=1, queries=300, n=100000, dim=128, nlists=256, nprobe=16)
    many = run(backends=16, inflight=16, queries=600, n=100000, dim=128, nlists=256, nprobe=16)
    assert one["mismatches"] == 0 and many["mismatches"] == 0
    assert many["avg_batch"] >= 32
    assert many["aggregate_queries_per_s"] >= 10 * one["aggregate_queries_per_s"], (one, many)


def _poll_once(L, s, dim, wait_us=20000):
    ids = (C.c_int * 8)()
    qbuf = np.zeros((8, dim), np.float32)
    st, npb, kk, cap = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    n = L.ndb_service_poll(s, 8, wait_us, 0, ids, qbuf.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(npb), C.byref(kk),
                           C.byref(cap))
    _last_nprobe[0] = npb.value
    return n, ids, kk.value


_last_nprobe = [None]


def test_a_scan_on_another_index_or_another_generation_is_refused_not_answered():
    """The request carries (index key, generation); the owner publishes what its mirror is.  Another key, or a
    generation the mirror does not hold, is NDBHIP_ERR_NODEVICE for the backend (its CPU scan takes over) and a
    newer generation asks the owner to reload — never rows from the wrong index (csrc/ndb_service.cpp)."""
    from neurondb_amd import _lib
    L = _lib.lib()
    dim = 8
    name = f"/ndb_service_ident_{os.getpid()}"
    s, c = C.c_void_p(), C.c_void_p()
    _lib.check(L.ndb_service_create(name.encode(), dim, 16, 4, C.byref(s)))
    try:
        _lib.check(L.ndb_service_publish(s, 16385, 7, 12))
        _lib.check(L.ndb_client_connect(name.encode(), C.byref(c)))
        k_, v_ = C.c_uint64(), C.c_uint64()
        _lib.check(L.ndb_client_index(c, C.byref(k_), C.byref(v_)))
        assert (k_.value, v_.value, L.ndb_client_meta_nprobe(c)) == (16385, 7, 12)
        q = np.ones(dim, np.float32)
        t = C.c_int(-1)
        nodev = -2
        assert L.ndb_client_submit_index(c, 999, 7, q.ctypes.data_as(C.c_void_p), 1, 2, 3, 0, C.byref(t)) == nodev
        assert b"another index" in L.ndbhip_last_error()
        assert L.ndb_client_submit(c, q.ctypes.data_as(C.c_void_p), 1, 2, 3, 0, C.byref(t)) == nodev   # unkeyed: key 0
        assert L.ndb_service_reload_wanted(s, None) == 0
        assert L.ndb_client_submit_index(c, 16385, 9, q.ctypes.data_as(C.c_void_p), 1, 2, 3, 0, C.byref(t)) == nodev
        w = C.c_uint64()
        assert L.ndb_service_reload_wanted(s, C.byref(w)) == 1 and w.value == 9
        # the matching request goes through ...
        _lib.check(L.ndb_client_submit_index(c, 16385, 7, q.ctypes.data_as(C.c_void_p), 1, 2, 3, 0, C.byref(t)))
        n, ids, kk = _poll_once(L, s, dim)
        assert n == 1 and kk == 3
        tids = np.zeros((1, 3, 6), np.uint8)
        dist = np.zeros((1, 3), np.float32)
        cnt = np.array([0], np.int32)
        _lib.check(L.ndb_service_complete(s, 1, ids, tids.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
                                          cnt.ctypes.data_as(C.c_void_p), 3, 0))
        got = C.c_int(-1)
        _lib.check(L.ndb_client_wait(c, t.value, None, None, C.byref(got), 1000))
        # ... and one submitted just before the owner moved to generation 9 is refused by the owner itself
        _lib.check(L.ndb_client_submit_index(c, 16385, 7, q.ctypes.data_as(C.c_void_p), 1, 2, 3, 0, C.byref(t)))
        _lib.check(L.ndb_service_publish(s, 16385, 9, 12))
        assert L.ndb_service_reload_wanted(s, None) == 0
        n, _, _ = _poll_once(L, s, dim, 1000)
        assert n == 0
        assert L.ndb_client_wait(c, t.value, None, None, C.byref(got), 1000) == nodev
        # the AM callback passes the scan's identity along
        _lib.check(L.ndb_am_use_service(name.encode()))
        assert not L.ndb_hnswbeginscan(None, 0, 1)               # hnsw scans are not served
        scan = L.ndb_ivfbeginscan_service(4242, 1, 0, 1)
        assert scan
        from tests.test_extract_vector import vector_datum
        from tests.test_gpu_am import VECTOR
        datum = vector_datum(q)
        buf = C.create_string_buffer(datum, len(datum))
        key = _lib.NdbScanKey(1, VECTOR, C.cast(buf, C.c_void_p), len(datum))
        _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(key), 1))
        assert L.ndb_ivfgettuple(scan, 1) == nodev
        L.ndb_ivfendscan(scan)
        # ref_compat: the index's own nprobe comes from the owner, not the default 10
        _lib.check(L.ndb_am_set_guc(b"neurondb.ref_compat", 1))
        scan = L.ndb_ivfbeginscan_service(16385, 9, 0, 1)
        _lib.check(L.ndb_ivfrescan(scan, None, 0, C.byref(key), 1))

        def one_poll():
            n, ids, kk = _poll_once(L, s, dim, 2_000_000)
            seen.append((n, kk))
            L.ndb_service_complete(s, n, ids, None, None, None, kk, 0)
        seen = []
        th = threading.Thread(target=one_poll)
        th.start()
        rc = L.ndb_ivfgettuple(scan, 1)
        assert rc == 0, L.ndbhip_last_error()
        th.join(10)
        L.ndb_ivfendscan(scan)
        assert seen == [(1, 10)] and _last_nprobe[0] == 12
    finally:
        L.ndb_am_set_guc(b"neurondb.ref_compat", 0)
        L.ndb_am_use_service(None)
        L.ndb_client_disconnect(c)
        L.ndb_service_destroy(s)


def _bump(name, key, times, q):
    from neurondb_amd import _lib
    L = _lib.lib()
    g = C.c_void_p()
    _lib.check(L.ndb_gen_attach(name.encode(), 64, C.byref(g)))
    seen = [L.ndb_gen_bump(g, key) for _ in range(times)]
    L.ndb_gen_detach(g, None)
    q.put(seen)


def test_index_generations_only_grow_and_are_shared_between_processes():
    """The mirrors' version stamp (include/ndb_service.h: ndb_gen_*): an insert followed by a vacuum must not bring
    an old stamp back (the reference's insertedVectors does, ivf_am.c:1346), and every backend must see every
    other backend's bump."""
    from neurondb_amd import _lib
    L = _lib.lib()
    name = f"/ndb_gen_test_{os.getpid()}"
    g = C.c_void_p()
    _lib.check(L.ndb_gen_attach(name.encode(), 64, C.byref(g)))
    try:
        assert L.ndb_gen_get(g, 0) == 0 and L.ndb_gen_bump(g, 0) == 0
        assert L.ndb_gen_get(g, 777) == 1                       # untouched index
        assert L.ndb_gen_bump(g, 777) == 2 and L.ndb_gen_get(g, 777) == 2
        assert L.ndb_gen_bump(g, 777) == 3                      # "insert, then vacuum": 2 -> 3, never back to 1
        assert L.ndb_gen_get(g, 778) == 1                       # another index is another counter
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_bump, args=(name, 900 + (r % 2), 500, q)) for r in range(4)]
        for p in procs:
            p.start()
        seen = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(60)
        for sq in seen:
            assert all(b > a for a, b in zip(sq, sq[1:]))        # each process sees its index's stamp move forward only
        assert L.ndb_gen_get(g, 900) == 1001 and L.ndb_gen_get(g, 901) == 1001   # no bump lost
        allv = sorted(v for sq in seen for v in sq)
        assert len(set(allv)) == 1000                            # the 2 x 1000 bumps: each value handed out once per key
        # a table that fills up says so instead of aliasing two indexes
        full = [L.ndb_gen_bump(g, 10_000 + i) for i in range(80)]
        assert full.count(0) >= 80 - 64 + 3 and all(v in (0, 2) for v in full)
        # ... and an index it could not take is of UNKNOWN generation (0: "reload for every scan"), not of generation 1 —
        # its changes can no longer be counted, so a mirror stamped 1 must never look fresh again; the indexes that got a
        # cell keep counting
        lost = [10_000 + i for i, v in enumerate(full) if v == 0]
        assert all(L.ndb_gen_get(g, k) == 0 for k in lost)
        assert L.ndb_gen_get(g, 777) == 3 and L.ndb_gen_bump(g, 777) == 4
    finally:
        L.ndb_gen_detach(g, name.encode())


def _claim_and_die(name, dim):
    from neurondb_amd import _lib
    L = _lib.lib()
    c = C.c_void_p()
    _lib.check(L.ndb_client_connect(name.encode(), C.byref(c)))
    q = np.zeros(dim, np.float32)
    t = C.c_int()
    for _ in range(3):
        _lib.check(L.ndb_client_submit(c, q.ctypes.data_as(C.c_void_p), 1, 1, 1, 0, C.byref(t)))
    os._exit(0)                                                  # a backend killed with requests in flight


def test_slots_of_dead_backends_are_reclaimed_and_a_dead_owner_is_noticed():
    from neurondb_amd import _lib
    L = _lib.lib()
    dim = 4
    name = f"/ndb_service_leak_{os.getpid()}"
    s, c = C.c_void_p(), C.c_void_p()
    _lib.check(L.ndb_service_create(name.encode(), dim, 2, 4, C.byref(s)))
    try:
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=_claim_and_die, args=(name, dim))
        p.start()
        p.join(60)
        _lib.check(L.ndb_client_connect(name.encode(), C.byref(c)))
        q = np.zeros(dim, np.float32)
        t = C.c_int()
        # 3 of 4 slots are READY for a dead pid: a 2nd live request would find no room ...
        _lib.check(L.ndb_client_submit(c, q.ctypes.data_as(C.c_void_p), 1, 1, 1, 0, C.byref(t)))   # the 4th slot
        # ... until the owner takes them back (its poll loop does so once a second; here directly)
        assert L.ndb_service_reclaim(s) == 3
        t2 = C.c_int()
        _lib.check(L.ndb_client_submit(c, q.ctypes.data_as(C.c_void_p), 1, 1, 1, 0, C.byref(t2)))
        n, ids, kk = _poll_once(L, s, dim)
        assert n == 2                                            # the dead backend's requests are not run
        assert t2.value != t.value
    finally:
        L.ndb_client_disconnect(c)
        L.ndb_service_destroy(s)
