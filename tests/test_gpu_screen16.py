"""The screened scan on fp16 matrix cores (csrc/ndbhip_screen16.h; scan mode 5, and auto mode from 32 queries):
results must be the oracle's bit for bit — ivfCollectCandidates, /root/reference/NeuronDB/src/index/ivf_am.c:1722-1909 —
and the statistics must show that this path (not a fallback) produced them."""
import numpy as np
import pytest

from tests.util import assert_same_results, make_ivf_arrays, oracle_image, oracle_search_batch

pytestmark = pytest.mark.gpu


def _index(a, nlists=None):
    from neurondb_amd import IvfIndex
    ix = IvfIndex(a["centroids"].shape[1], nlists if nlists is not None else len(a["list_len"]))
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    return ix


@pytest.fixture
def lib():
    from neurondb_amd import _lib
    _lib.ensure_init()
    yield _lib
    _lib.check(_lib.lib().ndbhip_set_scan_mode(0))
    _lib.check(_lib.lib().ndbhip_set_option(b"screen16", 1))
    _lib.check(_lib.lib().ndbhip_set_option(b"screen16_records", 8192))


@pytest.mark.parametrize("dim,n,nlists,nq", [(768, 6000, 24, 200), (100, 5000, 16, 150), (33, 4000, 9, 130),
                                               (128, 20000, 40, 300), (1536, 2500, 6, 128)])
def test_screen16_matches_oracle_and_really_runs(dim, n, nlists, nq, lib):
    """Any dim (the planes are padded to 32), ragged lists and query tiles, k up to 64, candidate cap, L2 and
    inner product; auto mode picks this path from 32 queries."""
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 11, dup_frac=0.05, zero_rows=2, empty_lists=(1,))
    ix = _index(a)
    img = oracle_image(a)
    rng = np.random.default_rng(dim)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 4] = a["base"][rng.integers(0, n, nq // 4)]                  # exact hits
    q[nq // 4: nq // 2] = (a["base"][rng.integers(0, n, nq // 2 - nq // 4)] +
                           0.01 * rng.standard_normal((nq // 2 - nq // 4, dim))).astype(np.float32)
    q[-1] = 0.0
    for mode in (5, 0):
        lib.check(lib.lib().ndbhip_set_scan_mode(mode))
        for strategy in (1, 3):
            for nprobe, k, cap in ((8, 10, 0), (nlists, 64, 0), (3, 1, 0), (6, 10, 100)):
                lib.check(lib.lib().ndbhip_stats_reset())
                t, d, c = ix.search(q, strategy, nprobe, k, cap)
                st = lib.stats()
                et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
                assert_same_results(t, d, c, et, ed, ec)
                assert st["screen16_batches"] + st["screen16_fallbacks"] == 1, st
                assert st["rows_emitted"] >= st["rows_rescored"] > 0 or st["screen16_fallbacks"] == 1, st
    ix.close()


def test_screen16_keeps_most_candidates_out_of_the_second_pass(lib):
    """On clustered data (the bench's kind) the bound pass must exclude nearly everything: a handful of
    survivors per query, no fallback — otherwise the path is correct but pointless."""
    rng = np.random.default_rng(5)
    dim, n, nlists, nq = 256, 40000, 64, 512
    cen = rng.standard_normal((nlists, dim)).astype(np.float32)
    lab = rng.integers(0, nlists, n)
    base = (cen[lab] + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
    order = np.argsort(lab, kind="stable")
    from oracle import ndbo
    a = dict(centroids=cen, list_len=np.bincount(lab, minlength=nlists).astype(np.int64),
             rows=np.ascontiguousarray(base[order]), tids=ndbo.tids_from_rows(order))
    ix = _index(a)
    img = oracle_image(a)
    q = (cen[rng.integers(0, nlists, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    lib.check(lib.lib().ndbhip_stats_reset())
    t, d, c = ix.search(q, 1, 8, 10)
    st = lib.stats()
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 8, 10)
    assert_same_results(t, d, c, et, ed, ec)
    assert st["screen16_batches"] == 1 and st["screen16_fallbacks"] == 0, st
    per_query = st["rows_rescored"] / nq
    assert 10 <= per_query < 80, per_query             # of ~5000 candidates per query
    ix.close()


def test_a_nan_row_only_affects_the_queries_that_probe_it(lib):
    """a single NaN row used to poison the bound of EVERY query of a screened batch (its norm won the
    row-norm maximum).  Now a row with a non-finite norm is excluded from the maximum and always handed to the
    reference's arithmetic; queries that do not probe its list must equal the oracle, in every scan mode."""
    a = make_ivf_arrays(6000, 64, 12, seed=77)
    img_clean = oracle_image(a)
    rng = np.random.default_rng(78)
    nq = 160
    q = (a["base"][rng.integers(0, 6000, nq)] + 0.01 * rng.standard_normal((nq, 64))).astype(np.float32)
    nprobe, k = 3, 10
    # the list the FEWEST queries probe gets the NaN row; only queries that do not probe it are compared
    probes = np.stack([img_clean.select_clusters(qq, nprobe) for qq in q])
    counts = np.bincount(probes.ravel(), minlength=12)
    victim = int(np.argmin(np.where(a["list_len"] > 0, counts, 10 ** 9)))
    off = np.concatenate([[0], np.cumsum(a["list_len"])])
    rows = a["rows"].copy()
    rows[off[victim] + 1, 5] = np.nan
    rows[off[victim] + 2, 7] = np.inf
    b = dict(a, rows=rows)
    ix = _index(b)
    clean = ~(probes == victim).any(1)
    assert clean.sum() >= 100
    et, ed, ec, _ = oracle_search_batch(img_clean, q[clean], 1, nprobe, k)
    for mode in (5, 3, 2, 1, 0):
        lib.check(lib.lib().ndbhip_set_scan_mode(mode))
        t, d, c = ix.search(q, 1, nprobe, k)
        assert_same_results(t[clean], d[clean], c[clean], et, ed, ec)
    ix.close()


@pytest.mark.parametrize("strategy", [1, 3])
def test_rows_whose_norm_overflows_go_to_the_reference_arithmetic_and_equal_the_oracle(strategy, lib):
    """vector_in rejects NaN and Inf (/root/reference/NeuronDB/src/core/neurondb.c:396,427) but not 1e20: a finite row
    whose sum of squares overflows float4.  Its L2 distance is +inf for every query (ivf_am.c:1562-1568: sum += diff*diff
    overflows, sqrtf(inf)) — it orders after every finite candidate and is returned when a query has fewer than k of
    those; its inner product is finite and huge — it is the FIRST or the LAST neighbour of every query that probes its
    list.  The screens cannot bound such a row (its plane norm is not finite): it must be handed to the reference's
    arithmetic and the results must be the oracle's for EVERY query, in every scan mode."""
    a = make_ivf_arrays(6000, 64, 12, seed=177)
    rng = np.random.default_rng(178)
    nq, nprobe, k = 160, 3, 10
    q = (a["base"][rng.integers(0, 6000, nq)] + 0.01 * rng.standard_normal((nq, 64))).astype(np.float32)
    probes = np.stack([oracle_image(a).select_clusters(qq, nprobe) for qq in q])
    counts = np.bincount(probes.ravel(), minlength=12)
    busiest = int(np.argmax(counts))
    off = np.concatenate([[0], np.cumsum(a["list_len"])])
    rows = a["rows"].copy()
    rows[off[busiest] + 4, 5] = np.float32(1e20)
    rows[off[busiest] + 9, :] = np.float32(-3e19)
    b = dict(a, rows=rows)
    img = oracle_image(b)
    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k)
    hit = (probes == busiest).any(1)
    assert hit.sum() >= 30
    if strategy == 3:
        big = np.abs(ed) > 1e18
        assert big[hit].any(), "the overflowing rows must show up among the inner-product neighbours"
    ix = _index(b)
    for mode in (5, 3, 2, 1, 0):
        lib.check(lib.lib().ndbhip_set_scan_mode(mode))
        t, d, c = ix.search(q, strategy, nprobe, k)
        assert_same_results(t, d, c, et, ed, ec)
    # every probed row a candidate (k beyond the lists' lengths): the +inf distances are part of the answer
    kk = 64
    small = dict(b)
    lens = np.asarray(b["list_len"]).copy()
    keep = np.concatenate([np.arange(off[L], off[L] + min(int(lens[L]), 15)) for L in range(12)])
    small["rows"], small["tids"] = b["rows"][keep], b["tids"][keep]
    small["list_len"] = np.minimum(lens, 15)
    et, ed, ec, _ = oracle_search_batch(oracle_image(small), q, strategy, nprobe, kk)
    if strategy == 1:
        assert np.isinf(ed[hit]).any()
    ix2 = _index(small)
    for mode in (5, 0):
        lib.check(lib.lib().ndbhip_set_scan_mode(mode))
        t, d, c = ix2.search(q, strategy, nprobe, kk)
        assert_same_results(t, d, c, et, ed, ec)
    ix.close()
    ix2.close()


@pytest.mark.parametrize("centered", [0, 1])
def test_overflowing_queries_fall_back_and_stay_exact(lib, centered):
    """centered = 0 (the two-plane sweep over the rows as they are): rows 300 from the origin and 0.01 apart —
    |q|^2 + |x|^2 - 2 q.x cancels almost completely, nothing can be excluded.  centered = 1 (the default: rows and
    queries minus the list's centroid, which removes exactly that cancellation): the rows of a list are copies of
    one another, so every candidate ties.  Either way every query overflows its records -> the batch is rerun on
    the fp32 screen; still the oracle's bits."""
    rng = np.random.default_rng(3)
    dim, n, nlists, nq = 64, 6000, 10, 140
    center = rng.standard_normal(dim).astype(np.float32) * 300.0
    if centered:
        proto = (center + rng.standard_normal((nlists, dim)).astype(np.float32) * 1e-2).astype(np.float32)
        base = proto[rng.integers(0, nlists, n)].copy()
    else:
        base = (center + rng.standard_normal((n, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    q = (center + rng.standard_normal((nq, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    cent = base[rng.choice(n, nlists, replace=False)].copy()
    asg = ((base[:, None, :].astype(np.float64) - cent[None]) ** 2).sum(-1).argmin(1)
    order = np.argsort(asg, kind="stable")
    from oracle import ndbo
    a = dict(centroids=cent, list_len=np.bincount(asg, minlength=nlists).astype(np.int64),
             rows=np.ascontiguousarray(base[order]), tids=ndbo.tids_from_rows(order))
    ix = _index(a)
    img = oracle_image(a)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    lib.check(lib.lib().ndbhip_set_option(b"screen16_records", 256))
    lib.check(lib.lib().ndbhip_set_option(b"screen16_centered", centered))
    lib.check(lib.lib().ndbhip_stats_reset())
    try:
        t, d, c = ix.search(q, 1, nlists, 10)
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_centered", 1))
    st = lib.stats()
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nlists, 10)
    assert_same_results(t, d, c, et, ed, ec)
    assert st["screen16_fallbacks"] == 1 and st["screen16_batches"] == 0, st
    ix.close()


def test_screen16_on_a_sharded_mirror_merges_to_the_unsharded_result(lib):
    import torch
    from neurondb_amd.dist import ShardedSearchBuffers
    a = make_ivf_arrays(9000, 96, 16, seed=41, dup_frac=0.1)
    img = oracle_image(a)
    rng = np.random.default_rng(42)
    nq, k, nprobe, world = 192, 10, 6, 4
    q = (a["base"][rng.integers(0, 9000, nq)] + 0.05 * rng.standard_normal((nq, 96))).astype(np.float32)
    full = _index(a)
    dq = torch.from_numpy(q).cuda()
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    bufs = [ShardedSearchBuffers(nq, k, world, "cuda") for _ in range(world)]
    lib.check(lib.lib().ndbhip_stats_reset())
    for w in range(world):
        owned = (np.arange(16) % world == w).astype(np.uint8)
        sh = full.shard(owned)
        sh.search_partial_device(dq, bufs[w].cand, bufs[w].ncand, bufs[w].total, 1, nprobe, k)
        lib.check(lib.lib().ndbhip_synchronize())
        bufs[0].cand_all[w].copy_(bufs[w].cand)
        bufs[0].ncand_all[w].copy_(bufs[w].ncand)
        sh.close()
    assert lib.stats()["screen16_batches"] == world
    b = bufs[0]
    ot = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    od = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(nq, dtype=torch.int32, device="cuda")
    lib.check(lib.lib().ndbhip_merge_topk_device(b.cand_all.data_ptr(), b.ncand_all.data_ptr(), bufs[0].total.data_ptr(),
                                                 world, nq, k, 3 * k, ot.data_ptr(), od.data_ptr(), oc.data_ptr()))
    lib.check(lib.lib().ndbhip_synchronize())
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
    from oracle import ndbo
    assert np.array_equal(oc.cpu().numpy(), ec)
    assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
    assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    full.close()


def test_list_level_pruning_drops_pairs_and_changes_nothing(lib):
    """screen16_prune: a (query, list) pair whose |q - centroid| - list radius already exceeds the query's threshold
    is not swept.  Tight clusters far apart (most pairs go), a list holding one row (radius 0), a list whose rows
    surround the query's own cluster at the same distance (radius as large as its centroid distance: must stay),
    a NaN row (its list's radius is +inf: never pruned), a query sitting exactly on a centroid."""
    rng = np.random.default_rng(77)
    dim, nlists, per = 96, 24, 300
    cents = (rng.standard_normal((nlists, dim)) * 5).astype(np.float32)    # (far apart, yet norms small enough for a tight bound)
    rows, lens = [], []
    for L in range(nlists):
        n = 1 if L == 5 else (120 if L == 7 else per)     # (120 equidistant rows: fewer than the 256 survivors finalize holds)
        r = cents[L] + 0.2 * rng.standard_normal((n, dim)).astype(np.float32)
        if L == 7:                                  # a shell of radius 8 around centroid 7
            u = rng.standard_normal((n, dim)).astype(np.float32)
            r = (cents[L] + 8.0 * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
        rows.append(r.astype(np.float32))
        lens.append(n)
    rows = np.concatenate(rows)
    rows[per * 9 + 3, 2] = np.nan                   # one row of one list is not finite
    from oracle import ndbo
    a = dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))
    ix = _index(a)
    img = oracle_image(a)
    nq, k, nprobe = 200, 10, 12
    src = rng.integers(0, nlists, nq)
    q = (cents[src] + 0.2 * rng.standard_normal((nq, dim))).astype(np.float32)
    q[0] = cents[3]                                 # exactly a centroid
    q[1] = rows[per * 7 + 2]                        # a stored row (distance 0 to itself)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    got = {}
    try:
        for prune in (1, 0):
            lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", prune))
            lib.check(lib.lib().ndbhip_stats_reset())
            got[prune] = ix.search(q, 1, nprobe, k) + (lib.stats(),)
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", 1))
    # (a query that probes the list with the NaN row is outside the contract, include/ndbhip.h: the others must not notice it)
    off = np.concatenate([[0], np.cumsum(lens)])
    victim = int(np.searchsorted(off, per * 9 + 3, side="right") - 1)
    with np.errstate(all="ignore"):
        probes = np.stack([img.select_clusters(qq, nprobe) for qq in q])
        clean = ~(probes == victim).any(1)
        assert clean.sum() >= 60
        et, ed, ec, _ = oracle_search_batch(img, q[clean], 1, nprobe, k)
    for prune in (1, 0):
        t, d, c, st = got[prune]
        assert_same_results(t[clean], d[clean], c[clean], et, ed, ec)
        assert st["screen16_batches"] == 1 and st["screen16_fallbacks"] == 0, st
    assert got[0][3]["pairs_pruned"] == 0
    assert got[1][3]["pairs_pruned"] > nq * nprobe // 2, got[1][3]           # most probed lists are far clusters
    assert got[1][3]["rows_swept"] < got[0][3]["rows_swept"]
    ix.close()


@pytest.mark.parametrize("strategy,cap,nprobe,dim,rowtype", [
    (1, 0, 6, 64, "f32"), (1, 40, 6, 64, "f32"), (3, 0, 4, 64, "f32"), (1, 0, 14, 64, "f32"), (1, 0, 6, 100, "f32"),
    (1, 0, 5, 33, "f32"), (3, 40, 6, 64, "f32"), (3, 0, 14, 100, "f32"),
    (3, 0, 4, 64, "f16"), (1, 0, 6, 64, "f16"), (3, 40, 6, 128, "f16"), (3, 0, 5, 64, "f16sub"),
    (2, 0, 6, 64, "f32"), (2, 40, 5, 100, "f32"), (2, 0, 14, 33, "f32"), (2, 0, 6, 64, "f16"), (2, 0, 5, 128, "f16sub")])
def test_sublists_regroup_long_lists_and_change_nothing(strategy, cap, nprobe, dim, rowtype, lib):
    """screen16_sublists with the threshold lowered to 300 rows: lists that mix several tight clusters are regrouped
    inside the planes (and a list of unstructured rows is not), (query, probe) pairs expand to sublists, seeds come
    from the nearest sublist.  Results must be the oracle's with and without it: positions, the k*10 candidate cap
    and ties are defined on the rows' places in their lists, which the regrouping must not disturb."""
    rng = np.random.default_rng(31 + nprobe + dim)
    nlists = 14
    comp = (rng.standard_normal((60, dim)) * 4).astype(np.float32)          # 60 tight clusters ...
    rows, lens = [], []
    for L in range(nlists):
        if L == 3:
            r = rng.standard_normal((900, dim)).astype(np.float32) * 4        # ... one list without structure
        else:
            mine = rng.choice(60, 1 + L % 5, replace=False)                   # ... mixed 1 to 5 per list, interleaved
            n = 150 + 170 * len(mine)
            r = (comp[mine[rng.integers(0, len(mine), n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
            r[7] = r[3]                                                       # duplicates: ties by position
            r[n - 1] = r[3]
        rows.append(r)
        lens.append(len(r))
    rows = np.concatenate(rows)
    from oracle import ndbo
    half = None
    if rowtype != "f32":
        # a halfvec column: the rows ARE fp16 values (f16sub: some of them fp16 subnormals, which the reference decodes
        # 2^-10 too small, quirk Q20); the oracle sees them as the reference decodes them
        if rowtype == "f16sub":
            rows[::7, 3] = np.float32(3e-6)
        half = rows.astype(np.float16).view(np.uint16)
        Lo = ndbo.lib()
        lut = np.array([Lo.ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
        rows = lut[half]
    cents = np.stack([rows[sum(lens[:L]):sum(lens[:L + 1])].mean(0) for L in range(nlists)]).astype(np.float32)
    a = dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))
    img = oracle_image(a)
    nq, k = 180, 10

    def make_index():
        if half is None:
            return _index(a)
        from neurondb_amd import IvfIndex
        ix = IvfIndex(dim, nlists)
        ix.set_centroids(cents)
        ix.load_f16(a["list_len"], half, a["tids"])
        return ix

    q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    try:
        for sublists in (1, 0):
            lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", sublists))
            lib.check(lib.lib().ndbhip_set_option(b"screen16_sub_min", 300))
            ix = make_index()
            lib.check(lib.lib().ndbhip_stats_reset())
            t, d, c = ix.search(q, strategy, nprobe, k, cap)
            st = lib.stats()
            assert_same_results(t, d, c, et, ed, ec)
            assert st["screen16_batches"] + st["screen16_fallbacks"] == 1, st
            if sublists and cap == 0 and dim == 64:
                # many sublists of the probed lists are excluded: |q - c| - radius (L2), -(q.c) - |q| radius (inner product)
                # (cosine cannot exclude a list that is its own sublist — list 3 here —: its centre lives in the rows' space)
                # (the regrouping's mini-k-means adds with atomics: the share that survives moves by a few per cent from
                # run to run — 0.62 .. 0.69 for inner product at nprobe 4)
                num, den = {1: (2, 3), 2: (4, 5), 3: (3, 4)}[strategy]
                assert st["rows_swept"] < st["rows_scored"] * num // den, st
            if half is not None:
                ix.close()
                continue
            # the mirror changes: an append invalidates the planes, the next batch regroups again
            ix.append(2, rows[5] + np.float32(0.001), ndbo.tids_from_rows(np.asarray([len(rows)]))[0])
            t2, d2, c2 = ix.search(q[:140], strategy, nprobe, k, cap)
            b = dict(a)
            off = np.concatenate([[0], np.cumsum(lens)])
            b["rows"] = np.insert(rows, off[3], rows[5] + np.float32(0.001), axis=0)
            b["tids"] = np.insert(a["tids"], off[3], ndbo.tids_from_rows(np.asarray([len(rows)]))[0])
            b["list_len"] = a["list_len"].copy()
            b["list_len"][2] += 1
            et2, ed2, ec2, _ = oracle_search_batch(oracle_image(b), q[:140], strategy, nprobe, k, cap)
            assert_same_results(t2, d2, c2, et2, ed2, ec2)
            ix.close()
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", 1))
        lib.check(lib.lib().ndbhip_set_option(b"screen16_sub_min", 256))


@pytest.mark.parametrize("strategy", [1, 2])
def test_a_query_that_keeps_more_sublists_than_the_count_pass_remembers(strategy, lib):
    """k_sub_pairs' count pass remembers up to 256 (sublist, probe) pairs per query for the fill pass to replay; a
    query that keeps more (40 probes x ~13 sublists here, nothing excluded with screen16_prune 0) switches the whole
    batch back to the fill pass that tests everything again.  Same results either way, and with pruning on."""
    dim, nlists, nprobe, nq, k = 32, 40, 40, 48, 10
    rng = np.random.default_rng(77)
    comp = (rng.standard_normal((nlists * 13, dim)) * 4).astype(np.float32)
    rows, lens = [], []
    for L in range(nlists):
        mine = np.arange(L * 13, L * 13 + 13)
        n = 1700
        rows.append((comp[mine[rng.integers(0, 13, n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32))
        lens.append(n)
    rows = np.concatenate(rows)
    from oracle import ndbo
    cents = np.stack([rows[sum(lens[:L]):sum(lens[:L + 1])].mean(0) for L in range(nlists)]).astype(np.float32)
    a = dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))
    img = oracle_image(a)
    q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, 0)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    try:
        swept, allst = [], []
        for prune in (0, 1):
            lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", prune))
            ix = _index(a)
            lib.check(lib.lib().ndbhip_stats_reset())
            t, d, c = ix.search(q, strategy, nprobe, k, 0)
            st = lib.stats()
            assert_same_results(t, d, c, et, ed, ec)
            assert st["screen16_batches"] == 1, st
            swept.append(st["rows_swept"])
            allst.append(st)
            ix.close()
        # every (query, sublist) pair kept without pruning; how much pruning then excludes is not this test's subject
        # (test_sublists_regroup_long_lists_and_change_nothing measures that on lists a regrouping is sure to help)
        assert swept[0] == nq * len(rows) and swept[1] <= swept[0], (swept, allst)
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", 1))


def test_sublists_on_slice_shards_merge_to_the_unsharded_result(lib):
    import torch
    from neurondb_amd.dist import ShardedSearchBuffers
    rng = np.random.default_rng(91)
    dim, nlists, world = 64, 6, 3
    comp = (rng.standard_normal((30, dim)) * 4).astype(np.float32)
    rows, lens = [], []
    for L in range(nlists):
        mine = rng.choice(30, 4, replace=False)
        n = 1200
        rows.append((comp[mine[rng.integers(0, 4, n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32))
        lens.append(n)
    rows = np.concatenate(rows)
    cents = np.stack([rows[1200 * L:1200 * (L + 1)].mean(0) for L in range(nlists)]).astype(np.float32)
    from oracle import ndbo
    from neurondb_amd.dist import partition_slices
    a = dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))
    img = oracle_image(a)
    nq, k, nprobe = 150, 10, 4
    q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    lib.check(lib.lib().ndbhip_set_option(b"screen16_sub_min", 300))
    try:
        full = _index(a)
        lo, ln, tail = partition_slices(a["list_len"], world, None, split_frac=0.0, align=16)    # every list cut into 3 slices
        shards = [full.shard_slices(lo[r], ln[r], tail[r]) for r in range(world)]
        dq = torch.from_numpy(q).cuda()
        bufs = [ShardedSearchBuffers(nq, k, world, "cuda") for _ in range(world)]
        for r in range(world):
            shards[r].search_partial_device(dq, bufs[r].cand, bufs[r].ncand, bufs[r].total, 1, nprobe, k)
            lib.check(lib.lib().ndbhip_synchronize())
            bufs[0].cand_all[r].copy_(bufs[r].cand)
            bufs[0].ncand_all[r].copy_(bufs[r].ncand)
        b0 = bufs[0]
        ot = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
        od = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
        oc = torch.zeros(nq, dtype=torch.int32, device="cuda")
        lib.check(lib.lib().ndbhip_merge_topk_device(b0.cand_all.data_ptr(), b0.ncand_all.data_ptr(), b0.total.data_ptr(),
                                                     world, nq, k, 3 * k, ot.data_ptr(), od.data_ptr(), oc.data_ptr()))
        lib.check(lib.lib().ndbhip_synchronize())
        assert np.array_equal(oc.cpu().numpy(), ec)
        assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
        assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))
        for sh in shards:
            sh.close()
        full.close()
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_sub_min", 256))


@pytest.mark.parametrize("dim,nlists,integer", [(64, 300, False), (100, 1100, False), (64, 700, True), (128, 2500, False)])
def test_centroid_scan_on_the_matrix_cores_selects_like_the_reference(dim, nlists, integer, lib):
    """ivfSelectClusters for a screened batch (k_cent_select, 256 .. 4096 centroids): |q - centroid|^2 from the
    matrix-core sweep, the reference's arithmetic only near the nprobe-th, the selection over those.  Ties (integer
    data), duplicated centroids (the lower index must win), a NaN and an overflowing centroid (never < FLT_MAX),
    nprobe beyond the centroids (everything probed: the old selection serves it), candidate cap."""
    n = 6 * nlists
    a = make_ivf_arrays(n, dim, nlists, seed=nlists, dup_frac=0.05, integer=integer)
    rng = np.random.default_rng(nlists + 1)
    cent = a["centroids"]
    cent[7] = cent[3]                       # duplicates: the first wins a tie
    cent[nlists - 1] = cent[nlists // 2]
    cent[11, 5] = np.nan
    cent[13, 2] = 3.0e38                    # the sum of squares overflows: +inf, never selected
    img = oracle_image(a)
    ix = _index(a)
    nq = 160
    q = rng.standard_normal((nq, dim)).astype(np.float32) if not integer else \
        rng.integers(-3, 4, size=(nq, dim)).astype(np.float32)
    q[:40] = cent[rng.integers(0, nlists, 40)]
    q[:40] = np.where(np.isfinite(q[:40]) & (np.abs(q[:40]) < 1e30), q[:40], 0.0)
    q[-1] = 0.0
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    for nprobe, k, cap in ((7, 10, 0), (40, 10, 0), (1, 3, 0), (16, 10, 300)):
        lib.check(lib.lib().ndbhip_stats_reset())
        t, d, c = ix.search(q, 1, nprobe, k, cap)
        st = lib.stats()
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k, cap)
        assert_same_results(t, d, c, et, ed, ec)
        assert st["cent_screen_batches"] == 1 and st["screen16_batches"] + st["screen16_fallbacks"] == 1, st
    # the probes themselves, and with the switch off
    from neurondb_amd import _lib as L
    lib.check(lib.lib().ndbhip_set_option(b"cent_screen16", 0))
    try:
        lib.check(lib.lib().ndbhip_stats_reset())
        t0, d0, c0 = ix.search(q, 1, 9, 10, 0)
        assert lib.stats()["cent_screen_batches"] == 0
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"cent_screen16", 1))
    t1, d1, c1 = ix.search(q, 1, 9, 10, 0)
    assert np.array_equal(c0, c1) and np.array_equal(np.asarray(t0).view(np.uint8), np.asarray(t1).view(np.uint8)) and \
        np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    if nlists <= 1024:
        # every centroid probed (nprobe >= centroids, also beyond nlists: list 0 again, ivf_am.c:1978)
        for nprobe in (nlists, nlists + 3):
            t, d, c = ix.search(q[:130], 1, nprobe, 10, 0)
            et, ed, ec, _ = oracle_search_batch(img, q[:130], 1, nprobe, 10, 0)
            assert_same_results(t, d, c, et, ed, ec)
    ix.close()


@pytest.mark.parametrize("sublists", [1, 0])
def test_appends_go_into_the_planes_spare_blocks_without_a_new_layout(sublists, lib):
    """aminsert between batches (ivf_am.c:954-1157): the centred planes keep spare blocks behind every list (behind an
    extra sublist around the centroid where the list is regrouped), so 1000 rows appended 25 at a time between
    130-query batches cost 40 small updates (stats.prepare_updates) and at most a couple of full layouts
    (stats.prepares: one at the start, one more only when a list outgrows its spare blocks) — and every batch is the
    oracle's, rows far from everything and duplicates of existing rows included."""
    from oracle import ndbo
    rng = np.random.default_rng(404 + sublists)
    dim, nlists, n0 = 64, 12, 20000
    cen = rng.standard_normal((40, dim)).astype(np.float32) * 3
    base = (cen[rng.integers(0, 40, n0)] + 0.1 * rng.standard_normal((n0, dim))).astype(np.float32)
    cent = base[rng.choice(n0, nlists, replace=False)].copy()
    asg = ((base[:, None, :].astype(np.float64) - cent[None]) ** 2).sum(-1).argmin(1)
    lists = [list(np.flatnonzero(asg == c)) for c in range(nlists)]
    rows_all = [base[i] for i in range(n0)]

    def arrays():
        order = np.concatenate([np.asarray(l, np.int64) for l in lists])
        return dict(centroids=cent, list_len=np.asarray([len(l) for l in lists], np.int64),
                    rows=np.ascontiguousarray(np.stack([rows_all[i] for i in order])), tids=ndbo.tids_from_rows(order))

    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", sublists))
    try:
        a = arrays()
        ix = _index(a)
        lib.check(lib.lib().ndbhip_stats_reset())
        nq, k, nprobe = 130, 10, 5
        for rnd in range(40):
            for _ in range(25):
                c = int(rng.integers(0, nlists))
                kind = rng.random()
                if kind < 0.8:
                    v = (cen[rng.integers(0, 40)] + 0.1 * rng.standard_normal(dim)).astype(np.float32)
                elif kind < 0.9:
                    v = rows_all[int(rng.integers(0, len(rows_all)))].copy()            # a duplicate
                else:
                    v = (rng.standard_normal(dim) * 30).astype(np.float32)              # far from everything
                rid = len(rows_all)
                rows_all.append(v)
                lists[c].append(rid)
                ix.append(c, v, ndbo.tids_from_rows(np.asarray([rid]))[0])
            q = (cen[rng.integers(0, 40, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
            q[:20] = np.stack([rows_all[i] for i in rng.integers(0, len(rows_all), 20)])
            t, d, c_ = ix.search(q, 1, nprobe, k, 0)
            if rnd % 8 == 7 or rnd < 2:
                et, ed, ec, _ = oracle_search_batch(oracle_image(arrays()), q, 1, nprobe, k, 0)
                assert_same_results(t, d, c_, et, ed, ec)
        st = lib.stats()
        assert st["screen16_batches"] + st["screen16_fallbacks"] == 40, st
        assert st["prepares"] <= 3 and st["prepare_updates"] >= 37, st
        ix.close()
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", 1))


@pytest.mark.parametrize("sublists", [1, 0])
def test_deletes_leave_holes_in_the_planes_without_a_new_layout(sublists, lib):
    """ambulkdelete between batches (ivf_am.c:1172-1357): the mirror is compacted (survivors keep list and order), the
    centred planes keep the deleted rows as holes and every surviving row's index in its list is renumbered from the
    compaction's own prefix sums — no new layout (stats.prepares stays 1), every batch the oracle's over the pruned
    arrays; VACUUMs that empty a whole list, hit duplicates' first copy, and appends in between included."""
    from oracle import ndbo
    rng = np.random.default_rng(505 + sublists)
    dim, nlists, n0 = 64, 10, 16000
    cen = rng.standard_normal((30, dim)).astype(np.float32) * 3
    base = (cen[rng.integers(0, 30, n0)] + 0.1 * rng.standard_normal((n0, dim))).astype(np.float32)
    base[101] = base[100]                                                              # duplicates: ties by position
    base[205] = base[100]
    cent = base[rng.choice(n0, nlists, replace=False)].copy()
    asg = ((base[:, None, :].astype(np.float64) - cent[None]) ** 2).sum(-1).argmin(1)
    asg[101] = asg[205] = asg[100]
    lists = [list(np.flatnonzero(asg == c)) for c in range(nlists)]
    rows_all = [base[i] for i in range(n0)]

    def arrays():
        order = np.concatenate([np.asarray(l, np.int64) for l in lists])
        return dict(centroids=cent, list_len=np.asarray([len(l) for l in lists], np.int64),
                    rows=np.ascontiguousarray(np.stack([rows_all[i] for i in order])), tids=ndbo.tids_from_rows(order))

    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", sublists))
    try:
        ix = _index(arrays())
        lib.check(lib.lib().ndbhip_stats_reset())
        nq, k, nprobe = 130, 10, 5
        smallest = int(np.argmin([len(l) for l in lists]))
        for rnd in range(12):
            if rnd > 0:
                live = [i for l in lists for i in l]
                if rnd == 3:
                    dead = list(lists[smallest])                                       # a VACUUM that empties a list
                elif rnd == 5:
                    dead = [100] if 100 in live else []                                # the first of three equal rows
                else:
                    dead = [int(i) for i in rng.choice(live, size=min(len(live) // 20, 700), replace=False)]
                dset = set(dead)
                for c in range(nlists):
                    lists[c] = [i for i in lists[c] if i not in dset]
                removed = ix.delete(ndbo.tids_from_rows(np.asarray(dead + [10 ** 7], np.int64)))   # (one TID that is not there)
                assert removed == len(dead)
                for _ in range(10):                                                    # ... and a few inserts behind them
                    c = int(rng.integers(0, nlists))
                    v = (cen[rng.integers(0, 30)] + 0.1 * rng.standard_normal(dim)).astype(np.float32)
                    rid = len(rows_all)
                    rows_all.append(v)
                    lists[c].append(rid)
                    ix.append(c, v, ndbo.tids_from_rows(np.asarray([rid]))[0])
            q = (cen[rng.integers(0, 30, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
            q[:20] = np.stack([rows_all[i] for i in rng.integers(0, n0, 20)])           # deleted rows among the queries
            q[20] = rows_all[100]
            t, d, c_ = ix.search(q, 1, nprobe, k, 0)
            et, ed, ec, _ = oracle_search_batch(oracle_image(arrays()), q, 1, nprobe, k, 0)
            assert_same_results(t, d, c_, et, ed, ec)
        st = lib.stats()
        assert st["screen16_batches"] + st["screen16_fallbacks"] == 12, st
        assert st["prepares"] == 1 and st["prepare_updates"] >= 20, st
        ix.close()
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_sublists", 1))


@pytest.mark.parametrize("strategy", [1, 3, 2])
def test_a_few_queries_with_too_many_ties_go_to_the_exact_path_alone(strategy, lib):
    """400 identical rows: a query next to them has 400 candidates no bound can separate — more survivors than the
    finalize stage holds.  Such queries (a handful per batch) are served by the exact path as a sub-batch of their own
    and put back; the other queries' results from the sweep stand, and the batch does not count as a fallback."""
    from oracle import ndbo
    rng = np.random.default_rng(77 + strategy)
    dim, nlists = 64, 8
    cen = (rng.standard_normal((24, dim)) * 4).astype(np.float32)
    base = (cen[rng.integers(0, 24, 6000)] + 0.05 * rng.standard_normal((6000, dim))).astype(np.float32)
    dup = (cen[5] + 0.05 * rng.standard_normal(dim)).astype(np.float32)
    base[1000:1400] = dup                                                            # 400 equal rows, one list
    cent = base[rng.choice(6000, nlists, replace=False)].copy()
    asg = ((base[:, None, :].astype(np.float64) - cent[None]) ** 2).sum(-1).argmin(1)
    order = np.argsort(asg, kind="stable")
    a = dict(centroids=cent, list_len=np.bincount(asg, minlength=nlists).astype(np.int64), rows=np.ascontiguousarray(base[order]),
             tids=ndbo.tids_from_rows(order))
    img = oracle_image(a)
    nq, k, nprobe = 160, 10, 4
    q = (cen[rng.integers(0, 24, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    q[[3, 50, 51, 120]] = dup + np.float32(0.001) * rng.standard_normal((4, dim)).astype(np.float32)
    q[77] = dup
    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, 0)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    ix = _index(a)
    lib.check(lib.lib().ndbhip_stats_reset())
    t, d, c = ix.search(q, strategy, nprobe, k, 0)
    st = lib.stats()
    assert_same_results(t, d, c, et, ed, ec)
    assert st["screen16_batches"] == 1 and st["screen16_fallbacks"] == 0, st
    ix.close()


@pytest.mark.parametrize("dim,n,nlists,nq", [(768, 9000, 12, 700), (100, 12000, 7, 300), (33, 5000, 3, 530), (64, 3000, 2, 260)])
@pytest.mark.parametrize("variant", ["dense pfd=3", "dense pfd=0", "dense pfd=7", "sweep<8,2>"])
def test_dense_tile_of_the_centred_sweep_matches_the_oracle(dim, n, nlists, nq, variant, lib):
    """The 256 x 256 tile (csrc/ndbhip_screen16d.h: loader and prefetcher waves, the matrix pipe's own screen of its
    accumulator blocks) on buckets probed by hundreds of queries: ragged row and pair tiles, exact hits and duplicates,
    zero rows, a list whose rows sit 2^40 and one whose rows sit 2^-40 from its centre next to ordinary ones
    (exponents outside the range the screen instruction is proved for: those items must take the per-element test),
    the candidate cap.  ids, ranks and float4 bits of ivfCollectCandidates (ivf_am.c:1722-1909)."""
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 5, dup_frac=0.05, zero_rows=2)
    off = np.zeros(nlists + 1, dtype=np.int64)
    off[1:] = np.cumsum(a["list_len"])
    rows = a["rows"]
    if nlists >= 7:
        # list 1: every row 2^40 x its offset from the centroid; list 2: 2^-40 x (rows AT the centre, up to rounding)
        for L, f in ((1, 2.0 ** 40), (2, 2.0 ** -40)):
            c = a["centroids"][L]
            rows[off[L]:off[L + 1]] = (c + f * (rows[off[L]:off[L + 1]] - c)).astype(np.float32)
    ix = _index(a)
    img = oracle_image(a)
    rng = np.random.default_rng(dim + 1)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 4] = rows[rng.integers(0, n, nq // 4)]
    q[nq // 4: nq // 2] = (rows[rng.integers(off[nlists - 1], n, nq // 2 - nq // 4)] +
                           0.01 * rng.standard_normal((nq // 2 - nq // 4, dim))).astype(np.float32)
    q[-1] = 0.0
    L = lib.lib()
    lib.check(L.ndbhip_set_option(b"screen16c_qb", 8))
    lib.check(L.ndbhip_set_option(b"screen16c_dense", 0 if variant.startswith("sweep") else 1))
    lib.check(L.ndbhip_set_option(b"screen16c_pfd", int(variant.split("=")[1]) if "pfd" in variant else 0))
    try:
        for nprobe, k, cap in ((nlists, 10, 0), (max(1, nlists // 2), 64, 0), (nlists, 10, 100)):
            lib.check(L.ndbhip_stats_reset())
            t, d, c = ix.search(q, 1, nprobe, k, cap)
            st = lib.stats()
            et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k, cap)
            assert_same_results(t, d, c, et, ed, ec)
            assert st["screen16_batches"] + st["screen16_fallbacks"] == 1, st
    finally:
        lib.check(L.ndbhip_set_option(b"screen16c_qb", 0))
        lib.check(L.ndbhip_set_option(b"screen16c_dense", 1))
        lib.check(L.ndbhip_set_option(b"screen16c_pfd", 0))
        ix.close()


@pytest.mark.parametrize("dim,rowtype,strategy", [(100, "f32", 1), (768, "f32", 1), (68, "f32", 3), (1536, "f32", 2),
                                                   (64, "f16", 1), (192, "f16sub", 3), (1536, "f16", 3), (320, "f16", 2)])
def test_rows_streamed_through_lds_sum_like_rows_read_by_their_lane(dim, rowtype, strategy, lib):
    """screen16_stage: k_s16_finalize's survivors and k_cent_select's candidate centroids take the reference's sequential
    arithmetic from rows copied chunk by chunk into LDS (s16_exact_staged) instead of loaded by the lane that sums them.
    Same steps, same order: the oracle's results at every ring depth (2 = one chunk ahead, 13 = the deepest), with dims
    that end inside a chunk (fp16 mirrors: dim % 64 == 0, a chunk is 128), fp16 rows (with subnormals: quirk Q20), more survivors than one pass of 16 holds (k = 40),
    and more candidate centroids than one pass of 64."""
    rng = np.random.default_rng(dim + strategy)
    n, nlists, nq = 6000, 300, 96
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 5, dup_frac=0.05, zero_rows=2)
    from oracle import ndbo
    half = None
    if rowtype != "f32":
        rows = a["rows"]
        if rowtype == "f16sub":
            rows[::7, 3] = np.float32(3e-6)
        half = rows.astype(np.float16).view(np.uint16)
        Lo = ndbo.lib()
        lut = np.array([Lo.ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
        a["rows"] = lut[half]
    img = oracle_image(a)
    if half is None:
        ix = _index(a)
    else:
        from neurondb_amd import IvfIndex
        ix = IvfIndex(dim, nlists)
        ix.set_centroids(a["centroids"])
        ix.load_f16(a["list_len"], half, a["tids"])
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 2] = (a["rows"][rng.integers(0, n, nq // 2)] + 0.01 * rng.standard_normal((nq // 2, dim))).astype(np.float32)
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    try:
        for nprobe, k in ((8, 10), (150, 40)):
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, 0)
            for stage in (0, 1, 2, 13):
                lib.check(lib.lib().ndbhip_set_option(b"screen16_stage", stage))
                lib.check(lib.lib().ndbhip_stats_reset())
                t, d, c = ix.search(q, strategy, nprobe, k, 0)
                st = lib.stats()
                assert_same_results(t, d, c, et, ed, ec)
                assert st["screen16_batches"] == 1 and st["screen16_fallbacks"] == 0 and st["cent_screen_batches"] == 1, (stage, st)
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_stage", 1))
    ix.close()


@pytest.mark.parametrize("strategy", [1, 3])
def test_dense_tile_on_a_halfvec_mirror(strategy, lib):
    """A halfvec mirror under the 256 x 256 tile with the sample-seeded thresholds switched on: the sample is gathered from
    float4 rows only (a fuzz campaign found the gather reading fp16 rows as floats, past the end of the mirror), so an
    fp16 mirror must take the plain seeds — and give the oracle's results."""
    from neurondb_amd import IvfIndex
    from oracle import ndbo
    dim, n, nlists, nq = 64, 5200, 6, 300
    a = make_ivf_arrays(n, dim, nlists, seed=77, dup_frac=0.05, zero_rows=2)
    half = a["rows"].astype(np.float16).view(np.uint16)
    Lo = ndbo.lib()
    lut = np.array([Lo.ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
    a["rows"] = lut[half]
    img = oracle_image(a)
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(a["centroids"])
    ix.load_f16(a["list_len"], half, a["tids"])
    rng = np.random.default_rng(5)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 2] = (a["rows"][rng.integers(0, n, nq // 2)] + 0.01 * rng.standard_normal((nq // 2, dim))).astype(np.float32)
    L = lib.lib()
    lib.check(L.ndbhip_set_scan_mode(5))
    lib.check(L.ndbhip_set_option(b"screen16c_qb", 8))
    lib.check(L.ndbhip_set_option(b"screen16c_sample", 256))
    try:
        for nprobe, k in ((nlists, 10), (3, 37)):
            t, d, c = ix.search(q, strategy, nprobe, k, 0)
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, 0)
            assert_same_results(t, d, c, et, ed, ec)
    finally:
        lib.check(L.ndbhip_set_option(b"screen16c_qb", 0))
        lib.check(L.ndbhip_set_option(b"screen16c_sample", 2048))
        ix.close()


def test_centres_of_probed_lists_only_give_the_same_results_as_the_full_matrix():
    """Round 6 (ivf_s16_sub_distances_probed, k_subdist_lists): a table with many regrouped lists scores, per batch, the
    sublist centres of each query's PROBED lists (list-major, fp32) instead of multiplying every query by every centre on
    the matrix cores.  The values only steer thresholds and pruning: results must be the oracle's either way — 256 lists
    that mix clusters (regrouped from 300 rows up), L2 and inner product, a candidate cap, duplicate probes of list 0
    (nprobe > lists is not possible here; list 0 is probed by many queries) — and `sub_restricted` must show which ran."""
    from neurondb_amd import IvfIndex, _lib
    from oracle import ndbo
    L = _lib.lib()
    rng = np.random.default_rng(606)
    dim, nlists = 64, 256
    comp = (rng.standard_normal((900, dim)) * 4).astype(np.float32)
    rows, lens = [], []
    for li in range(nlists):
        mine = rng.choice(900, 2 + li % 3, replace=False)
        n = 330 + 40 * (li % 7)
        r = (comp[mine[rng.integers(0, len(mine), n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
        rows.append(r)
        lens.append(n)
    rows = np.concatenate(rows)
    cents = np.stack([rows[sum(lens[:li]):sum(lens[:li + 1])].mean(0) for li in range(nlists)]).astype(np.float32)
    a = dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))
    q = (comp[rng.integers(0, 900, 384)] + 0.05 * rng.standard_normal((384, dim))).astype(np.float32)
    img = oracle_image(a)
    _lib.check(L.ndbhip_set_option(b"screen16_sub_min", 300))
    _lib.check(L.ndbhip_set_scan_mode(5))
    try:
        for strategy, cap in ((1, 0), (3, 0), (1, 60)):
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, 12, 10, cap)
            for restrict, expect in ((1, True), (0, False)):
                _lib.check(L.ndbhip_set_option(b"screen16_sub_restrict", restrict))
                ix = IvfIndex(dim, nlists)
                ix.set_centroids(a["centroids"])
                ix.load(a["list_len"], a["rows"], a["tids"])
                ix.search(q, strategy, 12, 10, cap)          # (the first batch lays the planes out; the second runs on them)
                _lib.check(L.ndbhip_stats_reset())
                t, d, c = ix.search(q, strategy, 12, 10, cap)
                st = _lib.stats()
                assert st["screen16_batches"] >= 1 and st["screen16_fallbacks"] == 0
                assert (st["sub_restricted"] >= 1) == expect, (strategy, restrict, st)
                assert_same_results(t, d, c, et, ed, ec)
                ix.close()
    finally:
        _lib.check(L.ndbhip_set_option(b"screen16_sub_restrict", 0))
        _lib.check(L.ndbhip_set_option(b"screen16_sub_min", 2048))
        _lib.check(L.ndbhip_set_scan_mode(0))


def test_a_list_probed_twice_offers_its_sublists_under_two_probes_and_the_seeds_take_one(lib):
    """Round 6's fuzz campaign (seed 81, case ~6000): nlists = 1 and nprobe = 2, so the reference scans list 0 again for the
    second probe slot (ivf_am.c:1978) and, with a candidate cap of 500 over 2 x 373 rows, only 127 rows of the second pass
    are candidates.  The seed kernel of the centred sweep (float4 seeds: the cosine path's, `screen16c_plane_seeds` 0 for L2)
    found the nearest sublist under BOTH probes at the same distance; its lanes broke the tie differently, disagreed on which
    seed rows are candidates, and one query got a threshold of 0 and lost its only neighbour.  The fixture is that case."""
    import os
    from tools.fuzz_replay import load_case
    from neurondb_amd import _lib as L
    z, ix, q, k, nprobe, cap, strategy, opts = load_case(os.path.join(os.path.dirname(__file__), "golden", "fuzz_r6_dup_probe_seed.npz"))
    a = dict(centroids=z["centroids"], list_len=z["list_len"], rows=z["rows"], tids=z["tids"])
    et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k, cap)
    assert nprobe > len(a["list_len"]) and cap < 2 * len(a["rows"]) and ec.min() == 1
    from tools.fuzz_scan import DEFAULT_OPTIONS, reset_options
    try:
        for name, value in opts.items():
            L.check(L.lib().ndbhip_set_option(name.encode(), value))
        for mode in (5, 0, 3):
            L.check(L.lib().ndbhip_set_scan_mode(mode))
            t, d, c = ix.search(q, strategy, nprobe, k, cap)
            assert_same_results(t, d, c, et, ed, ec)
        # ... and cosine, whose seeds are always the float4 rows'
        et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, 2, nprobe, k, cap)
        L.check(L.lib().ndbhip_set_scan_mode(5))
        t, d, c = ix.search(q, 2, nprobe, k, cap)
        assert_same_results(t, d, c, et, ed, ec)
    finally:
        reset_options(L.lib(), L.check)
        ix.close()
