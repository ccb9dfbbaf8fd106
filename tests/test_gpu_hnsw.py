"""Parity of the HIP HNSW search (hnswSearch, hnsw_am.c:1545-2080) against the
CPU oracle on graphs built by the oracle's literal hnswInsertNode: returned
blocks, ranks, float4 distances and the number of distance evaluations."""
import numpy as np
import pytest

from oracle import ndbo

pytestmark = pytest.mark.gpu


def build_graph(n, dim, m, efc, seed, normalize=False, dup=False, integer=False):
    rng = np.random.default_rng(seed)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    if integer:                                     # exact ties, exact zero dot products, zero vectors
        vecs = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
        vecs[5] = 0.0
    if normalize:
        vecs /= np.linalg.norm(vecs, axis=1, keepdims=True).astype(np.float32)
    if dup:
        vecs[n // 2:] = vecs[: n - n // 2]          # exact duplicates => distance ties
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 4)
    L = ndbo.lib()
    for i in range(n):
        r = float(rng.uniform(1e-9, 1.0))
        g.insert(vecs[i], i, L.ndbo_hnsw_level_from_uniform(r, np.float32(0.36)))
    return g, vecs


def load(g):
    from neurondb_amd import HnswIndex
    a = g.arrays()
    ix = HnswIndex(a["dim"], a["m"])
    ix.load(a["vecs"], a["levels"], a["ncount"], a["nbrs"], a["tids"].astype(np.uint16).view(np.uint8).reshape(-1, 6),
            a["entry_point"], a["entry_level"])
    return ix, a


def check(g, ix, queries, strategy, ef, k):
    """Both search kernels — one wave per query in the reference's summation order, and the block-cooperative
    one (dim % 4 == 0) — against the oracle: blocks, ranks, float4 bits, evaluation counts."""
    from neurondb_amd import _lib
    exp = [g.search(q, strategy, ef, k) for q in queries]
    try:
        for mode in ((1, 2) if ix.dim % 4 == 0 else (1,)):
            _lib.check(_lib.lib().ndbhip_hnsw_set_search_mode(mode))
            ob, od, oc, ot, sc = ix.search(queries, strategy, ef, k)
            for i, (eb, ed, ns) in enumerate(exp):
                assert oc[i] == len(eb), (mode, i, oc[i], len(eb))
                assert np.array_equal(ob[i, :len(eb)], eb), (mode, i, ob[i, :len(eb)], eb, od[i, :len(eb)], ed)
                assert np.array_equal(od[i, :len(eb)].view(np.uint32), ed.view(np.uint32)), \
                    (mode, i, od[i, :len(eb)], ed)
                assert sc[i] == ns, (mode, i, sc[i], ns)
    finally:
        _lib.check(_lib.lib().ndbhip_hnsw_set_search_mode(0))


@pytest.mark.parametrize("n,dim,m,efc", [(600, 16, 4, 20), (1500, 32, 8, 40), (400, 768, 16, 32), (300, 6, 5, 16)])
def test_hnsw_search_matches_oracle(n, dim, m, efc):
    g, vecs = build_graph(n, dim, m, efc, seed=n + dim)
    ix, a = load(g)
    assert a["entry_level"] >= 1          # the greedy descent is exercised
    rng = np.random.default_rng(1)
    q = rng.standard_normal((12, dim)).astype(np.float32)
    q[:3] = vecs[:3]
    for strategy in (1, 2, 3):
        for ef, k in ((64, 10), (8, 3), (200, 200), (4, 10)):
            check(g, ix, q, strategy, ef, k)


def test_hnsw_search_on_small_integer_vectors():
    """Ties everywhere, dot products that are exactly 0 (the interval of the block-cooperative scorer then
    straddles +-0 and the row is redone in order), a zero row and a zero query (cosine's 2.0f)."""
    g, vecs = build_graph(900, 16, 8, 40, seed=21, integer=True)
    ix, a = load(g)
    rng = np.random.default_rng(22)
    q = rng.integers(-2, 3, size=(24, 16)).astype(np.float32)
    q[0] = 0.0
    q[1] = vecs[5]
    q[2] = vecs[17]
    for strategy in (1, 2, 3):
        for ef, k in ((64, 10), (16, 16)):
            check(g, ix, q, strategy, ef, k)


def test_hnsw_cosine_on_unit_vectors_c3_shape():
    """configs[2] shape in miniature: m=16, ef_search=64, k=10, cosine, unit-norm rows."""
    g, vecs = build_graph(1200, 96, 16, 48, seed=3, normalize=True)
    ix, _ = load(g)
    q = np.random.default_rng(2).standard_normal((16, 96)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    check(g, ix, q, 2, 64, 10)


def test_hnsw_duplicate_vectors_ties():
    g, vecs = build_graph(500, 8, 6, 24, seed=9, dup=True)
    ix, _ = load(g)
    q = vecs[:10].copy()
    for strategy in (1, 2, 3):
        check(g, ix, q, strategy, 32, 10)


def test_hnsw_single_node_and_bad_strategy():
    from neurondb_amd import HnswIndex, NdbHipError
    g, vecs = build_graph(1, 8, 4, 8, seed=1)
    ix, _ = load(g)
    check(g, ix, vecs[:1] + 1.0, 1, 64, 10)
    with pytest.raises(NdbHipError):          # hnsw_am.c:1339-1343: ERROR on strategy outside {1,2,3}
        ix.search(vecs[:1], 4, 64, 10)


def test_hnsw_scan_state_machine():
    from neurondb_amd import HnswScan
    g, vecs = build_graph(300, 16, 4, 16, seed=4)
    ix, a = load(g)
    scan = HnswScan(ix, ef_search=32, k=5)
    scan.rescan(vecs[7], strategy=1)
    got = []
    while scan.gettuple():
        got.append(scan.xs_heaptid.copy())
    eb, ed, _ = g.search(vecs[7], 1, 32, 5)
    exp = [tuple(a["tids"][b]) for b in eb]
    assert [tuple(int(x) for x in t.tolist()) for t in got] == [tuple(int(x) for x in e) for e in exp]


# (optimistic, batch_div, batch_max): the one-wave sequential kernel; the default schedule; and a schedule
# that batches as many walks as there are nodes, so that conflicts (and their in-order redo) are the rule
BUILD_MODES = {"sequential": (0, 64, 1024), "optimistic": (1, 64, 1024), "optimistic-greedy": (1, 1, 256),
               "optimistic-wave-commit": (2, 64, 1024), "optimistic-wave-commit-greedy": (2, 1, 256),
               "optimistic-sorted-commit": (3, 64, 1024), "optimistic-sorted-commit-greedy": (3, 1, 256)}


@pytest.fixture
def restore_build_mode():
    yield
    from neurondb_amd import HnswIndex
    HnswIndex.set_build_mode(True, 64, 1024)


@pytest.mark.parametrize("mode", list(BUILD_MODES))
@pytest.mark.parametrize("n,dim,m,efc", [(500, 16, 4, 20), (700, 64, 8, 40), (300, 768, 16, 64), (200, 6, 5, 16),
                                         (2500, 128, 16, 200)])     # the reference's default m / ef_construction
def test_hnsw_device_build_matches_oracle_graph(n, dim, m, efc, mode, restore_build_mode):
    """hnswbuild on the device vs the oracle's literal hnswInsertNode: identical neighbour arrays (all 16
    levels, incl. the out-of-node back-links of Q12/Q21), counts, entry point — whichever way the inserts
    are scheduled (k_hnsw_build, or k_hnsw_spec + k_hnsw_commit)."""
    from neurondb_amd import HnswIndex
    HnswIndex.set_build_mode(*BUILD_MODES[mode])
    rng = np.random.default_rng(n * 3 + dim)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    vecs[n // 2] = vecs[3]                                   # duplicate vector
    L = ndbo.lib()
    levels = np.array([L.ndbo_hnsw_level_from_uniform(float(r), np.float32(0.36))
                       for r in rng.uniform(1e-9, 1.0, n)], np.int32)
    levels[7] = 3                                            # make sure upper levels (and self-links) occur
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    for i in range(n):
        g.insert(vecs[i], i, int(levels[i]))
    a = g.arrays()
    ix = HnswIndex(dim, m)
    ix.build(vecs, ndbo.tids_from_rows(np.arange(n)), levels, efc)
    st = ix.build_stats()
    if mode == "sequential":
        assert st["walks"] == 0
    else:
        # one walk per insert and linked level: min(level, entry level before the insert) + 1
        ent, walks = -1, 0
        for lv in levels:
            walks += (min(int(lv), ent) + 1) if ent >= 0 else 0
            ent = max(ent, int(lv))
        assert st["walks"] == walks
        assert st["rounds"] >= st["batches"] >= 1 and st["overflowed"] == 0
        if mode.endswith("greedy"):
            assert st["redone"] > 0 and st["max_batch"] > 16    # stale walks were met and run again
    e = ix.export()
    assert e["nblocks"] == a["nblocks"]
    assert (e["entry_point"], e["entry_level"]) == (a["entry_point"], a["entry_level"])
    assert np.array_equal(e["levels"][1:], a["levels"][1:])
    assert np.array_equal(e["ncount"][1:], a["ncount"][1:])
    assert np.array_equal(e["nbrs"][1:], a["nbrs"][1:])
    # and the device-built (dense) graph answers queries like the oracle
    q = rng.standard_normal((8, dim)).astype(np.float32)
    for strategy in (1, 2, 3):
        check(g, ix, q, strategy, 32, 10)


@pytest.mark.parametrize("built_on_device", [False, True])
def test_hnsw_bulkdelete_matches_oracle(built_on_device, restore_build_mode):
    """ndbhip_hnsw_delete vs the oracle's literal hnswbulkdelete: same unlinking (counts, shifted lists), same
    entry point hand-over, dead nodes stay reachable through the links that still name them (the reference's
    search never tests the dead flag), second VACUUM of the same TIDs is a no-op — on a loaded (packed) graph
    and on a device-built (dense) one."""
    from neurondb_amd import HnswIndex
    n, dim, m, efc = 900, 32, 6, 30
    rng = np.random.default_rng(5)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    L = ndbo.lib()
    levels = np.array([L.ndbo_hnsw_level_from_uniform(float(r), np.float32(0.36))
                       for r in rng.uniform(1e-9, 1.0, n)], np.int32)
    levels[11] = 4
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    for i in range(n):
        g.insert(vecs[i], i, int(levels[i]))
    a = g.arrays()
    ix = HnswIndex(dim, m)
    if built_on_device:
        ix.build(vecs, ndbo.tids_from_rows(np.arange(n)), levels, efc)
    else:
        ix.load(a["vecs"], a["levels"], a["ncount"], a["nbrs"], a["tids"], a["entry_point"], a["entry_level"])
    rows = np.unique(np.concatenate([rng.choice(n, 120, replace=False), [a["entry_point"] - 1, 11]]))
    tids = ndbo.tids_from_rows(rows)
    ghost = ndbo.tids_from_rows(np.array([n + 50, n + 51]))
    victims = np.concatenate([tids, ghost])
    assert g.bulkdelete(victims) == len(rows)
    assert ix.delete(victims[rng.permutation(len(victims))]) == len(rows)
    assert g.bulkdelete(victims) == 0 and ix.delete(victims) == 0
    b, e = g.arrays(), ix.export()
    assert (e["entry_point"], e["entry_level"]) == (b["entry_point"], b["entry_level"])
    assert (b["entry_point"], b["entry_level"]) != (a["entry_point"], a["entry_level"])
    if built_on_device:
        assert np.array_equal(e["ncount"][1:], b["ncount"][1:])
        assert np.array_equal(e["nbrs"][1:], b["nbrs"][1:])
    else:       # slots above a node's own levels (the Q12/Q21 out-of-node writes) are not part of a packed image
        for blk in range(1, n + 1):
            lv = b["levels"][blk]
            assert np.array_equal(e["ncount"][blk, :lv + 1], b["ncount"][blk, :lv + 1])
            assert np.array_equal(e["nbrs"][blk, :lv + 1], b["nbrs"][blk, :lv + 1])
    q = rng.standard_normal((16, dim)).astype(np.float32)
    for strategy in (1, 2, 3):
        check(g, ix, q, strategy, 32, 10)


@pytest.mark.parametrize("mode", ["sequential", "optimistic"])
def test_hnsw_insert_on_top_of_a_built_graph(mode, restore_build_mode):
    """hnswinsert after hnswbuild: 600 rows built, then 250 more in one call, then single rows — the graph is
    the oracle's after the same 900 inserts; and on top of a LOADED (packed) graph the searchable part of the
    graph (levels <= node level) and the search results agree."""
    from neurondb_amd import HnswIndex
    HnswIndex.set_build_mode(*BUILD_MODES[mode])
    n, n0, dim, m, efc = 900, 600, 24, 6, 30
    rng = np.random.default_rng(8)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    L = ndbo.lib()
    levels = np.array([L.ndbo_hnsw_level_from_uniform(float(r), np.float32(0.36))
                       for r in rng.uniform(1e-9, 1.0, n)], np.int32)
    levels[700] = 5                                          # a later insert takes the entry point over
    tids = ndbo.tids_from_rows(np.arange(n))
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    for i in range(n0):
        g.insert(vecs[i], i, int(levels[i]))
    half = g.arrays()
    for i in range(n0, n):
        g.insert(vecs[i], i, int(levels[i]))
    a = g.arrays()
    q = rng.standard_normal((12, dim)).astype(np.float32)

    ix = HnswIndex(dim, m)
    ix.build(vecs[:n0], tids[:n0], levels[:n0], efc)
    ix.insert(vecs[n0:n0 + 250], tids[n0:n0 + 250], levels[n0:n0 + 250], efc)
    for i in range(n0 + 250, n):
        ix.insert(vecs[i], tids[i:i + 1], levels[i:i + 1], efc)
    e = ix.export()
    assert (e["nblocks"], e["entry_point"], e["entry_level"]) == (a["nblocks"], a["entry_point"], a["entry_level"])
    assert np.array_equal(e["levels"][1:], a["levels"][1:])
    assert np.array_equal(e["ncount"][1:], a["ncount"][1:])
    assert np.array_equal(e["nbrs"][1:], a["nbrs"][1:])
    for strategy in (1, 2):
        check(g, ix, q, strategy, 32, 10)

    ld = HnswIndex(dim, m)
    ld.load(half["vecs"], half["levels"], half["ncount"], half["nbrs"], half["tids"], half["entry_point"],
            half["entry_level"])
    ld.insert(vecs[n0:], tids[n0:], levels[n0:], efc)
    e = ld.export()
    assert (e["nblocks"], e["entry_point"], e["entry_level"]) == (a["nblocks"], a["entry_point"], a["entry_level"])
    for blk in range(1, n + 1):
        lv = a["levels"][blk]
        assert np.array_equal(e["ncount"][blk, :lv + 1], a["ncount"][blk, :lv + 1])
        assert np.array_equal(e["nbrs"][blk, :lv + 1], a["nbrs"][blk, :lv + 1])
    for strategy in (1, 3):
        check(g, ld, q, strategy, 32, 10)


def test_hnsw_relation_pages_round_trip():
    """index pages -> mirror -> search (== oracle on the same graph) -> VACUUM + inserts on the mirror ->
    pages again, read by the test's own reader: the image holds the mirror's graph incl. the dead flags."""
    from neurondb_amd import HnswIndex
    from tests import pgpages
    g, vecs = build_graph(500, 16, 6, 24, seed=12)
    a = g.arrays()
    t6 = np.ascontiguousarray(a["tids"]).view(np.uint8).reshape(-1, 6)
    img = pgpages.write_hnsw_reference_format(a["vecs"], a["levels"], a["ncount"], a["nbrs"], t6, a["entry_point"],
                                              a["entry_level"], 6, efc=24, efs=32)
    ix = HnswIndex.load_pages(img)
    assert (ix.dim, ix.m, ix.nblocks) == (16, 6, a["nblocks"])
    import ctypes as C
    from neurondb_amd import _lib
    efc, efs = C.c_int(), C.c_int()
    _lib.check(_lib.lib().ndbhip_hnsw_get_meta(ix._h, C.byref(efc), C.byref(efs)))
    assert (efc.value, efs.value) == (24, 32)               # HnswMetaPageData.efConstruction / efSearch
    # From here on the model is the oracle over what the PAGES hold: the reference's writes above a node's
    # own level (Q12/Q21) land outside the item — on a real page outside the buffer — so the image has no
    # such slots, while the oracle's in-memory build kept them and later inserts would read them back.
    r0 = pgpages.read_hnsw_image(img, 6)
    g = ndbo.HnswGraph.from_arrays(r0["vecs"], r0["levels"], r0["ncount"], r0["nbrs"], a["tids"],
                                   r0["entry_point"], r0["entry_level"], 6, 24, cap_nodes=600)
    rng = np.random.default_rng(13)
    q = rng.standard_normal((10, 16)).astype(np.float32)
    for strategy in (1, 2, 3):
        check(g, ix, q, strategy, 32, 10)
    victims = ndbo.tids_from_rows(np.array([4, 77, 300]))
    assert ix.delete(victims) == 3 and g.bulkdelete(victims) == 3
    more = rng.standard_normal((40, 16)).astype(np.float32)
    lv = np.zeros(40, np.int32)
    lv[5] = 2
    ix.insert(more, ndbo.tids_from_rows(np.arange(500, 540)), lv, 24)
    for i in range(40):
        g.insert(more[i], 500 + i, int(lv[i]))
    for strategy in (1, 2):
        check(g, ix, q, strategy, 32, 10)
    r = pgpages.read_hnsw_image(ix.write_pages(24, 32), 6)
    b = g.arrays()
    assert (r["entry_point"], r["entry_level"]) == (b["entry_point"], b["entry_level"])
    assert set(np.nonzero(r["dead"])[0].tolist()) == {5, 78, 301} and r["inserted"] == 540 - 3
    assert np.array_equal(r["vecs"][1:], b["vecs"][1:]) and np.array_equal(r["levels"][1:], b["levels"][1:])
    for blk in range(1, b["nblocks"]):
        lvl = b["levels"][blk]
        assert np.array_equal(r["ncount"][blk, :lvl + 1], b["ncount"][blk, :lvl + 1])
        assert np.array_equal(r["nbrs"][blk, :lvl + 1], b["nbrs"][blk, :lvl + 1])
    # and the image loads back into an equivalent mirror
    again = HnswIndex.load_pages(r and ix.write_pages(24, 32))
    for strategy in (1, 3):
        check(g, again, q, strategy, 32, 10)


def check_layer(g, ix, queries, ef, k):
    """hnsw_search_layer on the device against the oracle's restatement of src/scan/hnsw_scan.c: blocks in slot
    order, float4 bits, counts and the number of compute_l2_distance calls."""
    exp = [g.search_layer(q, ef, k) for q in queries]
    ob, od, oc, ot, sc = ix.search_layer(queries, ef, k)
    for i, (eb, ed, ns) in enumerate(exp):
        assert oc[i] == len(eb), (i, oc[i], len(eb))
        assert np.array_equal(ob[i, :len(eb)], eb), (i, ob[i, :len(eb)], eb, od[i, :len(eb)], ed)
        assert np.array_equal(od[i, :len(eb)].view(np.uint32), ed.view(np.uint32)), (i, od[i, :len(eb)], ed)
        assert sc[i] == ns, (i, sc[i], ns)
    return exp


def pages_model(g, cap_nodes=None):
    """The oracle graph a mirror loaded from PAGES stands for: a node's item ends after its own level's slots,
    so the reference's writes above that level (Q12 / Q21) are not in the image.  hnsw_scan.c reads layers
    without testing the node's level (:549), so — unlike hnswSearch — it can tell the difference."""
    a = g.arrays()
    nbrs = a["nbrs"].copy()
    for b in range(a["nblocks"]):
        nbrs[b, a["levels"][b] + 1:] = 0xFFFFFFFF
    return ndbo.HnswGraph.from_arrays(a["vecs"], a["levels"], a["ncount"], nbrs, a["tids"], a["entry_point"],
                                      a["entry_level"], a["m"], cap_nodes=cap_nodes)


@pytest.mark.parametrize("n,dim,m,efc", [(600, 16, 4, 20), (1500, 32, 8, 40), (400, 768, 16, 32), (300, 6, 5, 16),
                                         (700, 20, 40, 30)])
def test_hnsw_search_layer_matches_oracle(n, dim, m, efc):
    """the reference's unused best-first search (src/scan/hnsw_scan.c), rule for rule — heap of
    2 * ef with dropped inserts (ef = 1, 2), the results[k - 1] bound, replace-first-worst, slot order."""
    g, vecs = build_graph(n, dim, m, efc, seed=n + dim)
    ix, a = load(g)
    g = pages_model(g)
    rng = np.random.default_rng(5)
    q = rng.standard_normal((70, dim)).astype(np.float32)          # 70 queries: persistent blocks reuse bitmaps
    q[:3] = vecs[:3]
    for ef, k in ((64, 10), (8, 3), (1, 10), (2, 1), (200, 200), (16, 40)):
        check_layer(g, ix, q, ef, k)
    # the visited bitmaps were left clean: the same batch again gives the same answer
    check_layer(g, ix, q, 64, 10)
    # `strategy` is accepted and ignored (hnsw_scan.c:384): same rows for any value
    b1 = ix.search_layer(q[:5], 64, 10, strategy=1)
    b2 = ix.search_layer(q[:5], 64, 10, strategy=2)
    assert np.array_equal(b1[0], b2[0]) and np.array_equal(b1[1].view(np.uint32), b2[1].view(np.uint32))


def test_hnsw_search_layer_ties_duplicates_and_tiny_graphs():
    g, vecs = build_graph(900, 16, 8, 40, seed=21, integer=True)   # exact ties in the heap and in the results
    ix, _ = load(g)
    g = pages_model(g)
    q = np.random.default_rng(22).integers(-2, 3, size=(24, 16)).astype(np.float32)
    q[1] = vecs[5]
    for ef, k in ((64, 10), (4, 16)):
        check_layer(g, ix, q, ef, k)
    g, vecs = build_graph(500, 8, 6, 24, seed=9, dup=True)
    ix, _ = load(g)
    check_layer(pages_model(g), ix, vecs[:10].copy(), 32, 10)
    g, vecs = build_graph(1, 8, 4, 8, seed=1)
    ix, _ = load(g)
    exp = check_layer(g, ix, vecs[:1] + 1.0, 64, 10)
    assert len(exp[0][0]) == 1


def test_hnsw_search_layer_on_a_graph_loaded_from_pages_and_after_vacuum():
    from neurondb_amd import HnswIndex
    from tests import pgpages
    g, vecs = build_graph(500, 16, 6, 24, seed=12)
    a = g.arrays()
    t6 = np.ascontiguousarray(a["tids"]).view(np.uint8).reshape(-1, 6)
    img = pgpages.write_hnsw_reference_format(a["vecs"], a["levels"], a["ncount"], a["nbrs"], t6, a["entry_point"],
                                              a["entry_level"], 6, efc=24, efs=32)
    ix = HnswIndex.load_pages(img)                               # packed mirror: densified on first use
    r0 = pgpages.read_hnsw_image(img, 6)
    g = ndbo.HnswGraph.from_arrays(r0["vecs"], r0["levels"], r0["ncount"], r0["nbrs"], a["tids"],
                                   r0["entry_point"], r0["entry_level"], 6, 24, cap_nodes=600)
    q = np.random.default_rng(13).standard_normal((10, 16)).astype(np.float32)
    check_layer(g, ix, q, 32, 10)
    check(g, ix, q, 1, 32, 10)                                    # hnswSearch still agrees on the dense mirror
    victims = ndbo.tids_from_rows(np.array([4, 77, 300]))
    assert ix.delete(victims) == 3 and g.bulkdelete(victims) == 3
    check_layer(g, ix, q, 32, 10)


def test_hnsw_search_layer_on_a_device_built_graph(restore_build_mode):
    """A graph built on the device keeps the dense 16-level slots, out-of-node back-links included, exactly
    like the oracle's in-memory build: hnsw_search_layer reads them on both sides."""
    from neurondb_amd import HnswIndex
    n, dim, m, efc = 1200, 32, 8, 40
    rng = np.random.default_rng(77)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    L = ndbo.lib()
    levels = np.array([L.ndbo_hnsw_level_from_uniform(float(r), np.float32(0.36))
                       for r in rng.uniform(1e-9, 1.0, n)], np.int32)
    levels[7] = 3
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    for i in range(n):
        g.insert(vecs[i], i, int(levels[i]))
    ix = HnswIndex(dim, m)
    ix.build(vecs, ndbo.tids_from_rows(np.arange(n)), levels, efc)
    q = rng.standard_normal((40, dim)).astype(np.float32)
    for ef, k in ((64, 10), (8, 20)):
        check_layer(g, ix, q, ef, k)
