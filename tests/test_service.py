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
