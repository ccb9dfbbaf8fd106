"""Host form of the shard merge (ndbhip_merge_topk_host) against the oracle's
selection sort (ivf_am.c:1856-1881) on the full candidate array.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from neurondb_amd import _lib
from oracle import ndbo


def key_of(d):
    u = np.asarray(d, np.float32).view(np.uint32).copy()
    u[u == 0x80000000] = 0
    neg = (u & 0x80000000) != 0
    return np.where(neg, ~u, u | 0x80000000).astype(np.uint32)


def partial_records(dist, pos, tid, k):
    """What one rank emits: everything below its k-th value + its first 2k ties (by position)."""
    n = len(dist)
    if n == 0:
        return np.zeros(0, dtype=[("key", "<u4"), ("pos", "<u4"), ("tid", "<u8")])
    ky = key_of(dist)
    kk = min(k, n)
    T = np.sort(ky, kind="stable")[kk - 1]
    lt = np.nonzero(ky < T)[0]
    eq = np.nonzero(ky == T)[0]
    eq = eq[np.argsort(pos[eq], kind="stable")][: 2 * k]
    sel = np.concatenate([lt, eq])
    rec = np.zeros(len(sel), dtype=[("key", "<u4"), ("pos", "<u4"), ("tid", "<u8")])
    rec["key"] = np.asarray(dist, np.float32).view(np.uint32)[sel]
    rec["pos"] = pos[sel]
    rec["tid"] = tid[sel]
    return rec


def run_merge(dist, k, world, rng):
    n = len(dist)
    pos = np.arange(n, dtype=np.uint32)
    tid = (np.arange(n, dtype=np.uint64) * 7 + 1)
    owner = rng.integers(0, world, n)
    cap = 3 * k
    cand = np.zeros((world, 1, cap), dtype=[("key", "<u4"), ("pos", "<u4"), ("tid", "<u8")])
    ncand = np.zeros((world, 1), dtype=np.int32)
    for w in range(world):
        m = owner == w
        rec = partial_records(dist[m], pos[m], tid[m], k)
        assert len(rec) <= cap
        cand[w, 0, :len(rec)] = rec
        ncand[w, 0] = len(rec)
    total = np.array([n], dtype=np.int64)
    ot = np.zeros((1, k), dtype=np.uint64)
    od = np.zeros((1, k), dtype=np.float32)
    oc = np.zeros(1, dtype=np.int32)
    rc = _lib.lib().ndbhip_merge_topk_host(cand.ctypes.data, ncand.ctypes.data, total.ctypes.data, world, 1, k,
                                           cap, ot.ctypes.data, od.ctypes.data, oc.ctypes.data)
    assert rc == 0, _lib.last_error()
    order = np.zeros(max(k, 1), dtype=np.int64)
    cnt = ndbo.lib().ndbo_selection_topk(np.ascontiguousarray(dist, np.float32), n, k, order)
    assert oc[0] == cnt
    assert np.array_equal(ot[0, :cnt], tid[order[:cnt]]), (ot[0, :cnt], tid[order[:cnt]])
    assert np.array_equal(od[0, :cnt].view(np.uint32), dist[order[:cnt]].view(np.uint32))


@pytest.mark.parametrize("world", [1, 2, 8])
@pytest.mark.parametrize("k", [1, 3, 10])
def test_merge_matches_selection_sort_with_heavy_ties(world, k):
    rng = np.random.default_rng(100 * world + k)
    for trial in range(60):
        n = int(rng.integers(0, 200))
        # few distinct values => many ties, including inside the first k slots
        dist = rng.integers(0, 4, n).astype(np.float32) * np.float32(0.5)
        if trial % 3 == 0 and n:
            dist[:] = dist[0]            # everything ties
        run_merge(dist, k, world, rng)


def test_merge_random_floats_and_signed_zero():
    rng = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng.integers(1, 500))
        dist = rng.standard_normal(n).astype(np.float32)
        dist[rng.integers(0, n, 3)] = np.float32(-0.0)
        dist[rng.integers(0, n, 3)] = np.float32(0.0)
        run_merge(dist, 10, 4, rng)


def test_displacement_case_differs_from_plain_lexicographic_order():
    """[5, 5', 1]: the reference returns 1, 5', 5 — not 1, 5, 5' (swap moves slot 0 behind)."""
    dist = np.array([5.0, 5.0, 1.0], np.float32)
    order = np.zeros(3, dtype=np.int64)
    ndbo.lib().ndbo_selection_topk(dist, 3, 3, order)
    assert list(order) == [2, 1, 0]
    run_merge(dist, 3, 2, np.random.default_rng(0))
