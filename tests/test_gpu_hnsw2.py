"""The `intended` HNSW on the device (csrc/ndbhip_hnsw2.h) against its sequential definition, oracle/ndb_oracle_hnsw2.c:
the device-built graph equals the oracle's slot for slot under the same batch schedule, the device search returns
the oracle's blocks, float4 distances and evaluation counts, and the graph finds neighbours (recall@10 against a
float64 brute force) where the reference-compatible one does not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _levels(rng, n):
    r = rng.uniform(1e-12, 1.0, n)
    return np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)      # hnsw_am.c:1143-1161


def _data(kind, n, dim, nq, seed):
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        cen = rng.standard_normal((32, dim)).astype(np.float32)
        base = (cen[rng.integers(0, 32, n)] + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
        q = (cen[rng.integers(0, 32, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    elif kind == "integer":
        base = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)             # equal distances everywhere
        q = rng.integers(-2, 3, size=(nq, dim)).astype(np.float32)
    elif kind == "offset":
        # values around 100, where halves are 1/16 apart: the fp16 image of a row is a different point
        base = (100.0 + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
        q = (100.0 + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
    return base, q, _levels(rng, n)


@pytest.mark.parametrize("kind,n,dim,m,efc,bdiv,bmax,select", [
    ("normal", 1500, 48, 8, 40, 64, 1024, 1), ("clustered", 2500, 64, 16, 64, 16, 256, 1),
    ("integer", 1200, 40, 6, 32, 8, 64, 1), ("normal", 900, 100, 16, 200, 1, 1, 1),
    ("clustered", 2000, 64, 16, 64, 32, 512, 0), ("normal", 1100, 1100, 8, 24, 64, 1024, 1),
    ("clustered", 2500, 64, 16, 64, 16, 256, 3), ("normal", 1300, 48, 8, 40, 64, 1024, 2),
    ("clustered", 2500, 64, 16, 64, 16, 256, 5), ("clustered", 2200, 64, 8, 48, 32, 512, 7),
    ("integer", 1200, 40, 6, 32, 8, 64, 7)])
def test_intended_build_and_search_equal_the_oracle(kind, n, dim, m, efc, bdiv, bmax, select):
    from neurondb_amd import HnswIndex, _lib
    from oracle import ndbo
    base, q, levels = _data(kind, n, dim, 40, seed=n + dim)
    og = ndbo.HnswGraph(dim, m, efc, cap_nodes=n + 1)
    nb = og.build_intended(base, levels, batch_div=bdiv, batch_max=bmax, select=select)
    e = og.arrays()
    _lib.ensure_init()
    _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(select))
    try:
        ix = HnswIndex(dim, m)
        ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, efc, batch_div=bdiv, batch_max=bmax)
    finally:
        _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(1))
    d = ix.export()
    assert d["nblocks"] == n + 1 and d["entry_point"] == e["entry_point"] and d["entry_level"] == e["entry_level"]
    assert np.array_equal(d["levels"][1:], e["levels"][1:])
    assert np.array_equal(d["ncount"][1:], e["ncount"][1:]), np.argwhere(d["ncount"] != e["ncount"])[:5]
    assert np.array_equal(d["nbrs"][1:], e["nbrs"][1:]), np.argwhere(d["nbrs"] != e["nbrs"])[:5]
    if bmax == 1:
        assert nb == n
    # (ef <= 64: the result set lives sorted in the lanes' registers; above: in LDS — 1, 2, 63 / 64 / 65 are its edges)
    for ef, k in ((64, 10), (8, 8), (200, 37), (1, 1), (2, 2), (63, 10), (65, 10)):
        ob, od, oc, oe = ix.search_intended(q, ef, k)
        for i in range(len(q)):
            eb, ed, ns = og.search_intended(q[i], ef, k)
            assert oc[i] == len(eb) and np.array_equal(ob[i, :oc[i]], eb), (i, ob[i], eb)
            assert np.array_equal(od[i, :oc[i]].view(np.uint32), ed.view(np.uint32)) and oe[i] == ns, (i, oe[i], ns)
    ix.close()


@pytest.mark.parametrize("kind,n,dim,m,efc", [("clustered", 3000, 96, 8, 64), ("normal", 2500, 768, 16, 100), ("clustered", 4000, 128, 16, 200),
                                              ("normal", 1500, 1024, 8, 40), ("clustered", 2000, 20, 4, 32), ("offset", 2500, 64, 8, 48)])
def test_walk_on_fp16_rows_rescored_on_float4_equals_the_oracle(kind, n, dim, m, efc):
    """ndbhip_hnsw_search_intended_w16_device == ndbo_h2_search_w16 (oracle/ndb_oracle_hnsw2.c "WALK ROWS"): descent and
    layer search on the reference's float4_to_fp16 image of the rows (src/types/quantization.c:141-168; what
    hnsw_am.c:1436-1451 would read from a halfvec column), groups-of-four summation tree, the result set re-scored on the
    float4 rows — same blocks, same float4 distance bits, same evaluation counts; rows with values the encoder flushes
    (below 2^-14) or saturates (beyond 65504) included; the twin follows appended rows."""
    from neurondb_amd import HnswIndex, _lib
    from oracle import ndbo
    base, q, levels = _data(kind, n, dim, 40, seed=3 * n + dim)
    base[5, :4] = np.float32(3e-6)          # flushed to zero by the encoder
    base[9, 1] = np.float32(1e5)            # beyond the halves' range: +inf on the walk rows, finite on the float4 rows
    base[11] = base[10]                     # equal rows: ties broken by block number on both sides
    og = ndbo.HnswGraph(dim, m, efc, cap_nodes=n + 1)
    og.build_intended(base, levels, batch_div=16, batch_max=512, select=1)
    w16 = og.walk_rows()
    _lib.ensure_init()
    ix = HnswIndex(dim, m)
    ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, efc, batch_div=16, batch_max=512)
    differs = 0
    for ef, k in ((64, 10), (8, 8), (200, 37), (1, 1), (63, 10), (65, 10)):
        ob, od, oc, oe = ix.search_intended(q, ef, k, walk16=True)
        pb, pd, pc, pe = ix.search_intended(q, ef, k)
        for i in range(len(q)):
            eb, ed, ns = og.search_intended_w16(w16, q[i], ef, k)
            assert oc[i] == len(eb) and np.array_equal(ob[i, :oc[i]], eb), (ef, i, ob[i], eb)
            assert np.array_equal(od[i, :oc[i]].view(np.uint32), ed.view(np.uint32)) and oe[i] == ns, (ef, i, oe[i], ns)
            differs += int(oe[i] - max(ef, k) != pe[i] or not np.array_equal(ob[i], pb[i]))
    # where the halves are coarse the two walks are different walks (else this test could not tell them apart)
    assert differs > 0 or kind != "offset"
    ix.close()


@pytest.mark.parametrize("kind", ["normal", "clustered"])
def test_intended_graph_finds_the_neighbours(kind):
    """recall@10 against a float64 brute force at ef_search = 64 (m = 16, ef_construction = 200)"""
    from neurondb_amd import HnswIndex
    from oracle import ndbo
    n, dim, nq = 20000, 64, 200
    base, q, levels = _data(kind, n, dim, nq, seed=5)
    ix = HnswIndex(dim, 16)
    ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, 200)
    ob, od, oc, oe = ix.search_intended(q, 64, 10)
    d2 = ((q[:, None, :].astype(np.float64) - base[None].astype(np.float64)) ** 2).sum(-1)
    gt = np.argsort(d2, axis=1, kind="stable")[:, :10] + 1
    recall = np.mean([len(set(ob[i, :oc[i]].tolist()) & set(gt[i].tolist())) / 10 for i in range(nq)])
    assert recall >= (0.95 if kind == "clustered" else 0.7), recall
    # the walk on fp16 rows loses next to nothing (the returned distances are the float4 rows' either way)
    wb, wd, wc, we = ix.search_intended(q, 64, 10, walk16=True)
    recall16 = np.mean([len(set(wb[i, :wc[i]].tolist()) & set(gt[i].tolist())) / 10 for i in range(nq)])
    assert recall16 >= recall - 0.01, (recall, recall16)
    assert np.allclose(wd[3, :wc[3]], np.sqrt(d2[3, wb[3, :wc[3]].astype(np.int64) - 1]), rtol=1e-6)
    # the distances are the L2 distances of the blocks returned
    i = 3
    ex = np.sqrt(d2[i, ob[i, :oc[i]].astype(np.int64) - 1])
    assert np.allclose(od[i, :oc[i]], ex, rtol=1e-6)
    ix.close()


def test_two_batches_of_walks_in_flight_on_shares_of_one_graph():
    """ndbhip_hnsw_share: handles on the same graph with workspaces of their own, a thread and a stream each; what the
    lanes return is what the same batches return one after the other (and the oracle's, by the tests above); a shared graph
    is frozen."""
    import threading
    import torch
    from neurondb_amd import HnswIndex, _lib
    from neurondb_amd._lib import NdbHipError
    from oracle import ndbo
    n, dim, m = 6000, 96, 8
    base, q, levels = _data("clustered", n, dim, 600, seed=12)
    ix = HnswIndex(dim, m)
    ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, 64)
    batches = [q[i * 100:(i + 1) * 100] for i in range(6)]
    for w16 in (False, True):
        want = [ix.search_intended(b, 48, 10, walk16=w16) for b in batches]
        handles = [ix, ix.share()]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got, err = [None] * len(batches), []

        def lane(w):
            try:
                _lib.check(_lib.lib().ndbhip_set_thread_stream(streams[w].cuda_stream))
                with torch.cuda.stream(streams[w]):          # (the wrapper's own tensor work goes to the lane's stream too)
                    for rep in range(3):
                        for b in range(w, len(batches), 2):
                            got[b] = handles[w].search_intended(batches[b], 48, 10, walk16=w16)
                _lib.check(_lib.lib().ndbhip_set_thread_stream(None))
            except Exception as e:          # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=lane, args=(w,)) for w in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for g_, w_ in zip(got, want):
            for x, y in zip(g_, w_):
                assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
        with pytest.raises(NdbHipError):
            ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, 64)      # frozen
        # (the library refuses to destroy a graph with live shares; HnswIndex.close() closes its shares first)
        assert _lib.lib().ndbhip_hnsw_destroy(ix._h) < 0 and b"shares" in _lib.lib().ndbhip_last_error()
        if not w16:
            handles[1].close()
    ix.close()
    assert handles[1]._h is None



@pytest.mark.parametrize("kind,n,dim,m,efc", [("normal", 2500, 64, 8, 48), ("clustered", 3000, 96, 16, 64), ("scaled", 2500, 768, 16, 100),
                                              ("scaled", 1500, 1100, 8, 40), ("integer", 1500, 40, 6, 32), ("zeros", 1200, 32, 8, 40)])
def test_intended_search_under_the_operator_class_strategy_equals_the_oracle(kind, n, dim, m, efc):
    """ndbhip_hnsw_search_intended[_w16]_device(..., strategy, ...) == ndbo_h2_search_s for <-> / <=> / <#>
    (strategies 1 / 2 / 3) on rows that are NOT unit vectors — where L2 order, cosine order and inner-product order are three
    different orders ("scaled": every row times its own factor in 0.2 .. 5) — blocks, float4 distance bits (strategies 2, 3:
    hnswComputeDistance's own values, hnsw_am.c:1321-1337) and evaluation counts, on float4 rows and on fp16 walk rows."""
    from neurondb_amd import HnswIndex, _lib
    from oracle import ndbo
    src = {"scaled": "normal", "zeros": "normal"}.get(kind, kind)
    base, q, levels = _data(src, n, dim, 48, seed=3 * n + dim)
    rng = np.random.default_rng(n)
    if kind == "scaled":
        base = (base * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
        q = (q * rng.uniform(0.2, 5.0, (len(q), 1))).astype(np.float32)
    if kind == "zeros":                                                     # zero-norm rows and a zero query: cosine's 2.0 branch
        base[rng.integers(0, n, 40)] = 0.0
        q[5] = 0.0
    og = ndbo.HnswGraph(dim, m, efc, cap_nodes=n + 1)
    og.build_intended(base, levels, batch_div=16, batch_max=256, select=1)
    w16 = og.walk_rows() if dim % 4 == 0 and dim <= 1024 else None
    ix = HnswIndex(dim, m)
    ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, efc, batch_div=16, batch_max=256)
    differ = 0
    for strategy in (1, 2, 3):
        for ef, k in ((64, 10), (16, 16), (100, 20), (1, 1)):
            for walk16 in ([False, True] if w16 is not None else [False]):
                ob, od, oc, oe = ix.search_intended(q, ef, k, walk16=walk16, strategy=strategy)
                for i in range(len(q)):
                    eb, ed, ns = og.search_intended_s(q[i], strategy, ef, k, w16=w16 if walk16 else None)
                    assert oc[i] == len(eb) and np.array_equal(ob[i, :oc[i]], eb), (strategy, ef, walk16, i, ob[i], eb)
                    assert np.array_equal(od[i, :oc[i]].view(np.uint32), ed.view(np.uint32)) and oe[i] == ns, (strategy, ef, i, oe[i], ns)
                    # the returned values are hnswComputeDistance's for that pair
                    if strategy != 1 and oc[i]:
                        ref = np.array([ndbo.hnsw_distance(q[i], base[b - 1], strategy) for b in eb], np.float32)
                        assert np.array_equal(ref.view(np.uint32), od[i, :oc[i]].view(np.uint32))
        if strategy != 1 and kind == "scaled":
            b1 = ix.search_intended(q, 64, 10, strategy=1)[0]
            bs = ix.search_intended(q, 64, 10, strategy=strategy)[0]
            differ += int((b1 != bs).any(axis=1).sum())
    if kind == "scaled":
        assert differ > len(q)               # the three orders really are different orders here
    # strategy 1 through the new argument is the old entry point's answer
    a1 = ix.search_intended(q, 64, 10)
    a2 = ix.search_intended(q, 64, 10, strategy=1)
    assert all(np.array_equal(x, y) for x, y in zip(a1, a2))
    with pytest.raises(Exception):
        ix.search_intended(q, 64, 10, strategy=4)          # hnswComputeDistance's ERROR (:1339-1343)
    ix.close()


def test_device_fp64_sqrt_and_divide_are_correctly_rounded():
    """The cosine walk key is -dot * rinv(node) with rinv = 1 / sqrt(sum of squares) in fp64 on both sides (the device keeps
    it per node: k_h2_rinv): the device's fp64 sqrt and divide must be the IEEE results (they are expansions on gfx950, not
    instructions).  2^20 random operands through torch's kernels would not prove the library's; so the search itself is the
    witness: a table whose rows differ in the last bits of their norms."""
    from neurondb_amd import HnswIndex
    from oracle import ndbo
    rng = np.random.default_rng(5)
    n, dim = 3000, 32
    d0 = rng.standard_normal(dim).astype(np.float32)
    # rows = one direction at 3000 lengths one float32 ulp apart + tiny noise: cosine keys differ in their last fp64 bits
    base = np.stack([(d0 * np.float32(1.0 + i * 2.0 ** -20) + np.float32(1e-6) * rng.standard_normal(dim)).astype(np.float32) for i in range(n)])
    q = (d0[None, :] + 1e-3 * rng.standard_normal((32, dim))).astype(np.float32)
    levels = _levels(rng, n)
    og = ndbo.HnswGraph(dim, 8, 40, cap_nodes=n + 1)
    og.build_intended(base, levels, batch_div=16, batch_max=256, select=1)
    ix = HnswIndex(dim, 8)
    ix.build_intended(base, ndbo.tids_from_rows(np.arange(n)), levels, 40, batch_div=16, batch_max=256)
    ob, od, oc, oe = ix.search_intended(q, 64, 10, strategy=2)
    for i in range(len(q)):
        eb, ed, ns = og.search_intended_s(q[i], 2, 64, 10)
        assert np.array_equal(ob[i, :oc[i]], eb) and oe[i] == ns
    ix.close()


@pytest.mark.parametrize("kind,n0,dim,m,efc,select", [("clustered", 1500, 64, 8, 48, 1), ("normal", 0, 48, 8, 40, 1), ("integer", 900, 40, 6, 32, 7)])
def test_rows_appended_to_an_intended_graph_equal_the_oracle(kind, n0, dim, m, efc, select):
    """hnswinsert under `intended` (ndbhip_hnsw_insert_intended_device == ndbo_h2_build on a graph that is not empty): calls
    of 1, 7, 300 rows and the rest on top of n0 built rows, the schedule going on from the relation's size — graph slot for
    slot after every call, searches equal at the end; walk rows follow the appended rows."""
    from neurondb_amd import HnswIndex, _lib
    from oracle import ndbo
    n = n0 + 1 + 7 + 300 + 400
    base, q, levels = _data(kind, n, dim, 24, seed=n + dim)
    tids = ndbo.tids_from_rows(np.arange(n))
    og = ndbo.HnswGraph(dim, m, efc, cap_nodes=n + 1)
    _lib.ensure_init()
    _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(select))
    try:
        ix = HnswIndex(dim, m)
        lo = 0
        for cnt in ([n0] if n0 else []) + [1, 7, 300, 400]:
            og.build_intended(base[lo:lo + cnt], levels[lo:lo + cnt], tids=tids[lo:lo + cnt], batch_div=16, batch_max=256, select=select)
            ix.build_intended(base[lo:lo + cnt], tids[lo:lo + cnt], levels[lo:lo + cnt], efc, batch_div=16, batch_max=256, append=lo > 0)
            lo += cnt
            d, e = ix.export(), og.arrays()
            assert d["nblocks"] == lo + 1 == e["nblocks"] and (d["entry_point"], d["entry_level"]) == (e["entry_point"], e["entry_level"])
            assert np.array_equal(d["levels"][1:], e["levels"][1:])
            assert np.array_equal(d["ncount"][1:], e["ncount"][1:]), (lo, np.argwhere(d["ncount"] != e["ncount"])[:5])
            assert np.array_equal(d["nbrs"][1:], e["nbrs"][1:]), (lo, np.argwhere(d["nbrs"] != e["nbrs"])[:5])
            if lo == n0 + 8 and dim % 4 == 0:
                ix.search_intended(q, 32, 5, walk16=True)            # walk rows made here must be made again after the next append
    finally:
        _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(1))
    w16 = og.walk_rows() if dim % 4 == 0 else None
    for strategy in (1, 2):
        for walk16 in ([False, True] if w16 is not None else [False]):
            ob, od, oc, oe = ix.search_intended(q, 48, 10, walk16=walk16, strategy=strategy)
            for i in range(len(q)):
                eb, ed, ns = og.search_intended_s(q[i], strategy, 48, 10, w16=w16 if walk16 else None)
                assert oc[i] == len(eb) and np.array_equal(ob[i, :oc[i]], eb) and oe[i] == ns
                assert np.array_equal(od[i, :oc[i]].view(np.uint32), ed.view(np.uint32))
    ix.close()
