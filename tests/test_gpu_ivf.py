"""Parity of the HIP IVF path (through the C ABI) against the CPU oracle:
neighbour ids, ranks AND float4 distances bit-identical.  Runs on the MI355X."""
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


def _queries(a, nq, seed):
    rng = np.random.default_rng(seed)
    dim = a["base"].shape[1]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 4] = a["base"][rng.integers(0, len(a["base"]), nq // 4)]   # exact hits (distance 0)
    return q


def test_batch_distance_all_recipes_bitexact():
    """hot loops 1/2 in isolation: every rounding recipe, several dims (tiled, tail chunk, direct)."""
    import ctypes as C
    from neurondb_amd import _lib
    from oracle import ndbo
    _lib.ensure_init()
    L = ndbo.lib()
    rng = np.random.default_rng(3)
    for dim in (4, 28, 64, 100, 128, 768, 3, 130, 1536):
        nq, nv = 3, 200
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        v = rng.standard_normal((nv, dim)).astype(np.float32)
        v[5] = 0.0
        v[7] = q[0]
        for recipe, strategies in ((0, (1, 2, 3, 4)), (1, (1, 2, 3))):
            for s in strategies:
                out = np.zeros((nq, nv), np.float32)
                _lib.check(_lib.lib().ndbhip_batch_distance(q.ctypes.data, v.ctypes.data, out.ctypes.data,
                                                            nq, nv, dim, s, recipe))
                exp = np.zeros_like(out)
                for i in range(nq):
                    for j in range(nv):
                        if recipe == 0 and s == 4:
                            exp[i, j] = L.ndbo_ivf_l2sq(q[i], v[j], dim)
                        elif recipe == 0:
                            exp[i, j] = L.ndbo_ivf_distance(q[i], v[j], dim, s)
                        else:
                            exp[i, j] = L.ndbo_hnsw_distance(q[i], v[j], dim, s, None)
                assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (dim, recipe, s)


def test_batch_distance_operator_recipes_bitexact():
    """The SQL operators' own kernels on the device: scalar double (what a default x86-64
    build runs, incl. the Kahan L2 and the Q15 sign of <#>), the AVX2 / AVX-512 lane orders with their
    horizontal-sum tree and FMA cosine, and the halfvec operators (per-element fp16_to_float, Q20)."""
    from neurondb_amd import _lib
    from oracle import ndbo
    _lib.ensure_init()
    L = ndbo.lib()
    rng = np.random.default_rng(5)
    ops = {1: L.ndbo_op_l2, 2: L.ndbo_op_cosine, 3: L.ndbo_op_ip}
    hops = {1: L.ndbo_halfvec_l2, 2: L.ndbo_halfvec_cosine, 3: L.ndbo_halfvec_ip}
    for dim in (3, 7, 8, 15, 16, 28, 100, 768):
        nq, nv = 2, 130
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        v = (rng.standard_normal((nv, dim)) * rng.choice([1e-3, 1.0, 1e3], (nv, 1))).astype(np.float32)
        v[5] = 0.0
        v[7] = q[0]
        for recipe, simd in ((2, 0), (3, 8), (4, 16)):
            for s_ in (1, 2, 3):
                out = np.zeros((nq, nv), np.float32)
                _lib.check(_lib.lib().ndbhip_batch_distance(q.ctypes.data, v.ctypes.data, out.ctypes.data,
                                                            nq, nv, dim, s_, recipe))
                exp = np.array([[ops[s_](q[i], v[j], dim, simd) for j in range(nv)] for i in range(nq)], np.float32)
                assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (dim, recipe, s_)
        qh = (q * 0.5).astype(np.float16).view(np.uint16).copy()
        vh = np.clip(v, -6e4, 6e4).astype(np.float16).view(np.uint16).copy()
        vh[3, :min(dim, 4)] = [0x0001, 0x83FF, 0x0200, 0x8000][:min(dim, 4)]        # subnormals, -0
        for s_ in (1, 2, 3):
            out = np.zeros((nq, nv), np.float32)
            _lib.check(_lib.lib().ndbhip_batch_distance(qh.ctypes.data, vh.ctypes.data, out.ctypes.data,
                                                        nq, nv, dim, s_, 5))
            exp = np.array([[hops[s_](qh[i], vh[j], dim) for j in range(nv)] for i in range(nq)], np.float32)
            assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), (dim, "halfvec", s_)


@pytest.mark.parametrize("dim,n,nlists", [(4, 100, 10), (28, 1000, 10), (128, 10000, 100), (768, 4000, 64),
                                           (3, 300, 7), (100, 2000, 20)])
@pytest.mark.parametrize("strategy", [1, 2, 3])
def test_ivf_search_matches_oracle(dim, n, nlists, strategy):
    a = make_ivf_arrays(n, dim, nlists, seed=dim * 7 + nlists, dup_frac=0.05, zero_rows=2)
    ix = _index(a)
    img = oracle_image(a)
    q = _queries(a, 24, seed=dim)
    for nprobe, k in ((10, 10), (3, 5), (nlists, 10), (1, 1)):
        nprobe = min(nprobe, 64)
        t, d, c = ix.search(q, strategy, nprobe, k)
        et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k)
        assert_same_results(t, d, c, et, ed, ec)


def test_c1_config_oracle_built_index_ref_compat_and_intended():
    """BASELINE configs[0]: 10k x 128, lists=100, k=10 L2, index built by the oracle's k-means
    (reference build rule), searched in ref_compat (cap k*10, nprobe 10) and intended mode."""
    from oracle import ndbo
    rng = np.random.default_rng(2024)
    base = rng.standard_normal((10000, 128)).astype(np.float32)
    img, asg, iters = ndbo.build_ivf_image(base, 100, max_iter=8)
    a = dict(centroids=img.centroids, list_len=np.diff(img.list_off), rows=img.vecs, tids=img.tids, base=base)
    ix = _index(a)
    q = _queries(a, 32, seed=9)
    for cap in (100, 0):
        t, d, c = ix.search(q, 1, 10, 10, cap)
        et, ed, ec, _ = oracle_search_batch(img, q, 1, 10, 10, cap)
        assert_same_results(t, d, c, et, ed, ec)
    # centroid selection alone
    sel = ix.select_clusters(q, 10)
    for i in range(len(q)):
        assert np.array_equal(sel[i], img.select_clusters(q[i], 10))


def test_integer_data_massive_ties():
    """small integer coordinates => many equal distances, incl. inside the first k slots."""
    a = make_ivf_arrays(3000, 8, 12, seed=5, integer=True)
    ix = _index(a)
    img = oracle_image(a)
    rng = np.random.default_rng(1)
    q = rng.integers(-3, 4, size=(40, 8)).astype(np.float32)
    for strategy in (1, 2, 3):
        for k in (1, 10, 37):
            t, d, c = ix.search(q, strategy, 6, k)
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, 6, k)
            assert_same_results(t, d, c, et, ed, ec)


def test_all_rows_identical():
    n, dim = 700, 16
    a = make_ivf_arrays(n, dim, 5, seed=1)
    a["rows"][:] = a["rows"][0]
    ix = _index(a)
    img = oracle_image(a)
    q = np.stack([a["rows"][0], a["rows"][0] + 1.0]).astype(np.float32)
    t, d, c = ix.search(q, 1, 5, 10)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 5, 10)
    assert_same_results(t, d, c, et, ed, ec)


def test_edge_cases_empty_lists_small_index_and_nprobe_quirk():
    # empty lists among the probed ones
    a = make_ivf_arrays(500, 16, 8, seed=3, empty_lists=(0, 3, 7))
    assert a["list_len"][0] == 0
    ix = _index(a)
    img = oracle_image(a)
    q = _queries(a, 8, seed=4)
    t, d, c = ix.search(q, 1, 8, 10)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 8, 10)
    assert_same_results(t, d, c, et, ed, ec)
    # fewer candidates than k
    t, d, c = ix.search(q, 1, 1, 400)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 1, 400)
    assert_same_results(t, d, c, et, ed, ec)
    # nprobe > nlists: never-written probe slots re-scan list 0 (palloc0: ivf_am.c:1978)
    a2 = make_ivf_arrays(300, 8, 4, seed=8)
    ix2 = _index(a2)
    img2 = oracle_image(a2)
    q2 = _queries(a2, 6, seed=2)
    t, d, c = ix2.search(q2, 1, 7, 10)
    et, ed, ec, _ = oracle_search_batch(img2, q2, 1, 7, 10)
    assert_same_results(t, d, c, et, ed, ec)
    # completely empty index
    a3 = make_ivf_arrays(50, 8, 4, seed=8)
    from neurondb_amd import IvfIndex
    ix3 = IvfIndex(8, 4)
    ix3.set_centroids(a3["centroids"])
    ix3.load(np.zeros(4, np.int64), np.zeros((0, 8), np.float32), np.zeros((0, 6), np.uint8))
    t, d, c = ix3.search(q2, 1, 4, 10)
    assert (c == 0).all()


def test_scan_state_machine_mirrors_ivfgettuple():
    from neurondb_amd import IvfScan
    a = make_ivf_arrays(2000, 32, 20, seed=11)
    ix = _index(a)
    img = oracle_image(a)
    q = _queries(a, 3, seed=12)
    for ref_compat in (True, False):
        scan = IvfScan(ix, ref_compat=ref_compat)
        for qq in q:
            scan.rescan(qq, strategy=2, nprobe=4, k=7)
            got = []
            while scan.gettuple():
                got.append((scan.xs_heaptid.copy(), scan.xs_orderbyval))
            if ref_compat:
                et, ed, _ = img.search(qq, 1, 10, 10, 100)
            else:
                et, ed, _ = img.search(qq, 2, 4, 7, 0)
            assert len(got) == len(et)
            for (t, d), t2, d2 in zip(got, et, ed):
                assert t == t2 and np.float32(d).tobytes() == np.float32(d2).tobytes()
        scan.endscan()
        # wrong-dimension query: no rows, no error (ivf_am.c:1961-1972)
        scan.rescan(np.zeros(5, np.float32))
        assert scan.gettuple() is False


def test_sharded_partial_plus_merge_equals_single_device():
    """8-way list sharding emulated on one GPU: per-shard partial records -> merge == unsharded."""
    import torch
    from neurondb_amd import IvfIndex, _lib
    a = make_ivf_arrays(6000, 64, 32, seed=21, dup_frac=0.1)
    img = oracle_image(a)
    q = _queries(a, 16, seed=22)
    k, nprobe, world = 10, 8, 8
    cap = 3 * k
    off = np.zeros(33, np.int64)
    off[1:] = np.cumsum(a["list_len"])
    dq = torch.from_numpy(q).cuda()
    cand = torch.zeros((world, len(q), cap, 2), dtype=torch.int64, device="cuda")   # 16-byte records
    ncand = torch.zeros((world, len(q)), dtype=torch.int32, device="cuda")
    total = torch.zeros((world, len(q)), dtype=torch.int64, device="cuda")
    keep = []
    for w in range(world):
        owned = (np.arange(32) % world == w).astype(np.uint8)
        sel = np.concatenate([np.arange(off[l], off[l + 1]) for l in range(32) if owned[l]] or
                             [np.zeros(0, np.int64)]).astype(np.int64)
        ix = IvfIndex(64, 32)
        ix.set_centroids(a["centroids"])
        ix.load(a["list_len"], a["rows"][sel], a["tids"][sel], owned=owned)
        ix.search_partial_device(dq, cand[w], ncand[w], total[w], 1, nprobe, k)
        keep.append(ix)
    ot = torch.zeros((len(q), k), dtype=torch.int64, device="cuda")
    od = torch.zeros((len(q), k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(len(q), dtype=torch.int32, device="cuda")
    import ctypes as C
    _lib.check(_lib.lib().ndbhip_merge_topk_device(cand.data_ptr(), ncand.data_ptr(), total[0].data_ptr(),
                                                   world, len(q), k, cap, ot.data_ptr(), od.data_ptr(),
                                                   oc.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
    assert np.array_equal(oc.cpu().numpy(), ec)
    from oracle import ndbo
    assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
    assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    # the host merge agrees too
    hc, hn, ht = cand.cpu().numpy(), ncand.cpu().numpy(), total[0].cpu().numpy()
    ot2 = np.zeros((len(q), k), np.uint64)
    od2 = np.zeros((len(q), k), np.float32)
    oc2 = np.zeros(len(q), np.int32)
    _lib.check(_lib.lib().ndbhip_merge_topk_host(hc.ctypes.data, hn.ctypes.data, ht.ctypes.data, world, len(q),
                                                 k, cap, ot2.ctypes.data, od2.ctypes.data, oc2.ctypes.data))
    assert np.array_equal(ndbo.tids_from_device_u64(ot2), et) and np.array_equal(oc2, ec)


def test_sharded_search_with_query_split_selection():
    """The N > 1 flow of neurondb_amd.dist with cluster selection split by queries, 4 ranks emulated on one
    GPU: each shard selects for its slice (select_clusters_device), the slices are concatenated (= the
    all-gather), every shard scans its lists for all queries with the given probes, merge == unsharded."""
    import torch
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd.dist import query_slice
    a = make_ivf_arrays(5000, 64, 24, seed=31, dup_frac=0.1)
    img = oracle_image(a)
    q = _queries(a, 37, seed=32)                       # 37 queries: grouped scan, ragged last slice
    k, nprobe, world = 10, 6, 4
    cap = 3 * k
    off = np.zeros(25, np.int64)
    off[1:] = np.cumsum(a["list_len"])
    dq = torch.from_numpy(q).cuda()
    shards = []
    for w in range(world):
        owned = (np.arange(24) % world == w).astype(np.uint8)
        sel = np.concatenate([np.arange(off[l], off[l + 1]) for l in range(24) if owned[l]]).astype(np.int64)
        ix = IvfIndex(64, 24)
        ix.set_centroids(a["centroids"])
        ix.load(a["list_len"], a["rows"][sel], a["tids"][sel], owned=owned)
        shards.append(ix)
    s = query_slice(len(q), world, 0)[2]
    probes_all = torch.zeros((world * s, nprobe), dtype=torch.int32, device="cuda")
    for w in range(world):
        lo, hi, _ = query_slice(len(q), world, w)
        if hi > lo:
            shards[w].select_clusters_device(dq[lo:hi], probes_all[w * s:w * s + hi - lo], nprobe)
    _lib.check(_lib.lib().ndbhip_synchronize())
    probes = probes_all[:len(q)].contiguous()
    exp_probes = np.stack([img.select_clusters(qq, nprobe) for qq in q])
    assert np.array_equal(probes.cpu().numpy(), exp_probes)
    cand = torch.zeros((world, len(q), cap, 2), dtype=torch.int64, device="cuda")
    ncand = torch.zeros((world, len(q)), dtype=torch.int32, device="cuda")
    total = torch.zeros((world, len(q)), dtype=torch.int64, device="cuda")
    for w in range(world):
        shards[w].search_partial_probes_device(dq, probes, cand[w], ncand[w], total[w], 1, nprobe, k)
    ot = torch.zeros((len(q), k), dtype=torch.int64, device="cuda")
    od = torch.zeros((len(q), k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(len(q), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ndbhip_merge_topk_device(cand.data_ptr(), ncand.data_ptr(), total[0].data_ptr(),
                                                   world, len(q), k, cap, ot.data_ptr(), od.data_ptr(),
                                                   oc.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
    from oracle import ndbo
    assert np.array_equal(oc.cpu().numpy(), ec)
    assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
    assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    assert np.array_equal(total.cpu().numpy(), np.broadcast_to(total[0].cpu().numpy(), (world, len(q))))


@pytest.fixture
def scan_mode():
    from neurondb_amd import _lib
    _lib.ensure_init()

    def set_mode(m):
        _lib.check(_lib.lib().ndbhip_set_scan_mode(m))
    yield set_mode
    set_mode(0)


@pytest.mark.parametrize("dim,n,nlists", [(64, 3000, 12), (128, 6000, 40), (768, 3000, 24)])
def test_grouped_scan_is_bit_identical_to_per_query_scan_and_oracle(dim, n, nlists, scan_mode):
    """k_ivf_scan_grouped (query-grouped, packed fp32 pairs) vs k_ivf_scan vs the oracle."""
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 1, dup_frac=0.05, zero_rows=1)
    ix = _index(a)
    img = oracle_image(a)
    q = _queries(a, 150, seed=dim + 2)          # > 64 queries: several groups per list, short tails
    for strategy in (1, 3, 2):                  # cosine falls back to the per-query kernel
        for nprobe, k, cap in ((8, 10, 0), (3, 7, 0), (10, 10, 100)):
            res = {}
            for mode in (1, 2, 3, 5):
                scan_mode(mode)
                res[mode] = ix.search(q, strategy, nprobe, k, cap)
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
            for mode in (1, 2, 3, 5):
                assert_same_results(*res[mode], et, ed, ec)


def test_grouped_scan_sharded(scan_mode):
    import torch
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd.dist import ShardedSearchBuffers, gather_and_merge
    a = make_ivf_arrays(5000, 64, 16, seed=31, dup_frac=0.1)
    img = oracle_image(a)
    q = _queries(a, 100, seed=32)
    k, nprobe, world = 10, 6, 4
    full = _index(a)
    dq = torch.from_numpy(q).cuda()
    et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
    for mode in (1, 2, 3, 5):
        scan_mode(mode)
        bufs = [ShardedSearchBuffers(len(q), k, world, "cuda") for _ in range(world)]
        for w in range(world):
            owned = (np.arange(16) % world == w).astype(np.uint8)
            sh = full.shard(owned)
            sh.search_partial_device(dq, bufs[w].cand, bufs[w].ncand, bufs[w].total, 1, nprobe, k)
            _lib.check(_lib.lib().ndbhip_synchronize())
            bufs[0].cand_all[w].copy_(bufs[w].cand)
            bufs[0].ncand_all[w].copy_(bufs[w].ncand)
            sh.close()
        b = bufs[0]
        _lib.check(_lib.lib().ndbhip_merge_topk_device(b.cand_all.data_ptr(), b.ncand_all.data_ptr(),
                                                       b.total.data_ptr(), world, len(q), k, b.cap,
                                                       b.out_tids.data_ptr(), b.out_dist.data_ptr(),
                                                       b.out_count.data_ptr()))
        _lib.check(_lib.lib().ndbhip_synchronize())
        from oracle import ndbo
        assert np.array_equal(b.out_count.cpu().numpy(), ec)
        assert np.array_equal(ndbo.tids_from_device_u64(b.out_tids.cpu().numpy()), et)
        assert np.array_equal(b.out_dist.cpu().numpy().view(np.uint32), ed.view(np.uint32))


def test_mirror_from_reference_format_pages_and_back():
    """pages (reference on-disk format, with dead entries) -> device mirror -> search == oracle on the
    live rows; mirror -> pages -> unpack round trip."""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    from oracle import ndbo
    from tests import pgpages
    rng = np.random.default_rng(42)
    dim, nl = 28, 9
    cents = rng.standard_normal((nl, dim)).astype(np.float32)
    sizes = [50, 0, 120, 33, 7, 260, 1, 90, 14]
    lists, r = [], 0
    for L in range(nl):
        ent = []
        for _ in range(sizes[L]):
            t = ndbo.tids_from_rows(np.array([r]))[0]
            ent.append((rng.standard_normal(dim).astype(np.float32), t.tobytes()))
            r += 1
        lists.append(ent)
    dead = {(5, 0), (5, 100), (2, 119), (0, 10)}
    img = pgpages.write_reference_format(cents, lists, dead=dead)
    a = np.frombuffer(img, np.uint8).copy()
    h = C.c_void_p()
    _lib.ensure_init()
    _lib.check(_lib.lib().ndbhip_ivf_load_pages(C.byref(h), a.ctypes.data, len(a) // 8192))
    ix = IvfIndex.__new__(IvfIndex)
    ix.dim, ix.nlists, ix._h, ix._keep, ix.ncent = dim, nl, h, [], nl
    # oracle image of the live rows
    rows, tids, ll = [], [], []
    for L in range(nl):
        keep = [(v, t) for i, (v, t) in enumerate(lists[L]) if (L, i) not in dead]
        ll.append(len(keep))
        rows += [v for v, _ in keep]
        tids += [np.frombuffer(t, ndbo.TID_DTYPE)[0] for _, t in keep]
    off = np.zeros(nl + 1, np.int64)
    off[1:] = np.cumsum(ll)
    oimg = ndbo.IvfImage(cents, off, np.array(rows, np.float32), np.array(tids, ndbo.TID_DTYPE))
    q = rng.standard_normal((12, dim)).astype(np.float32)
    t, d, c = ix.search(q, 1, 4, 10)
    et, ed, ec, _ = oracle_search_batch(oimg, q, 1, 4, 10)
    assert_same_results(t, d, c, et, ed, ec)
    # back to pages
    need = _lib.lib().ndbhip_ivf_pages_needed(dim, nl, np.array(ll, np.int64).ctypes.data)
    out = np.zeros(need * 8192, np.uint8)
    nb = C.c_uint32()
    _lib.check(_lib.lib().ndbhip_ivf_write_pages(ix._h, 10, out.ctypes.data, need, C.byref(nb)))
    d2, c2, ll2, rows2, t62, ver = pgpages.read_image(out[: nb.value * 8192].tobytes())
    assert ver == 1 and list(ll2) == ll
    assert np.array_equal(rows2.view(np.uint32), np.array(rows, np.float32).view(np.uint32))


def test_large_k_and_nprobe_limits():
    """k and nprobe up to the GUC maxima (hnsw_k / ivf_probes: 1000) take the radix path and the big LDS carve."""
    a = make_ivf_arrays(9000, 16, 300, seed=77, dup_frac=0.05)
    ix = _index(a)
    img = oracle_image(a)
    q = _queries(a, 6, seed=78)
    for nprobe, k in ((300, 10), (64, 200), (250, 1000), (1000, 65)):
        t, d, c = ix.search(q, 1, nprobe, k)
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
        assert_same_results(t, d, c, et, ed, ec)
    from neurondb_amd import NdbHipError
    with pytest.raises(NdbHipError):
        ix.search(q, 1, 10, 2000)
    with pytest.raises(NdbHipError):
        ix.search(q, 1, 0, 10)


@pytest.mark.parametrize("dim,n,nlists", [(64, 3000, 10), (128, 5000, 20), (1536, 1200, 8)])
def test_fp16_rows_are_bit_identical_to_expanded_float4_rows(dim, n, nlists, scan_mode):
    """halfvec column kept as fp16 in HBM: decode on the fly == the reference's fp16_to_float (incl. the
    subnormal quirk Q20), so every result equals the search over the expanded float4 rows."""
    from neurondb_amd import IvfIndex
    from oracle import ndbo
    L = ndbo.lib()
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 5, dup_frac=0.05)
    h = (a["rows"] * 0.25).astype(np.float16).view(np.uint16).copy()
    h[3, :8] = [0x0001, 0x0200, 0x03FF, 0x8001, 0x83FF, 0x0000, 0x8000, 0x0400]     # subnormals, signed zeros
    lut = np.array([L.ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
    rows32 = lut[h]                                   # what the reference indexes for these halfvec values
    img = oracle_image(dict(a, rows=rows32))
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(a["centroids"])
    ix.load_f16(a["list_len"], h, a["tids"])
    q = _queries(a, 100, seed=dim)
    q[0] = rows32[3]
    for mode in (1, 2, 3, 5):
        scan_mode(mode)
        for strategy in (3, 1, 2):                    # config 5 is inner product
            t, d, c = ix.search(q, strategy, 5, 10)
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, 5, 10)
            assert_same_results(t, d, c, et, ed, ec)


@pytest.mark.parametrize("reference_encoder", [True, False])
def test_to_f16_twin_and_its_shards(reference_encoder, scan_mode):
    """ndbhip_ivf_to_f16: rows narrowed on the device with the reference's float4_to_fp16 (truncating,
    flush-to-zero) or round-to-nearest-even; the twin — and its list shards — search like the oracle over
    the re-expanded rows."""
    import torch
    from neurondb_amd import IvfIndex, _lib
    from oracle import ndbo
    L = ndbo.lib()
    dim, n, nlists = 128, 4000, 16
    a = make_ivf_arrays(n, dim, nlists, seed=77, dup_frac=0.05)
    rows = (a["rows"] * 0.37).astype(np.float32)
    # flush-to-zero and the smallest-normal edge; values that overflow fp16 (-> inf, then inf - inf = NaN in a
    # score) are left out: a NaN distance is outside the parity contract ("Non-finite distances")
    rows[5, :6] = [1e-6, -1e-6, 65000.0, -65000.0, 6.1e-5, 5.9e-5]
    a = dict(a, rows=rows)
    if reference_encoder:
        h = np.array([L.ndbo_float4_to_fp16(float(v)) for v in rows.reshape(-1)], np.uint16).reshape(rows.shape)
    else:
        with np.errstate(over="ignore"):
            h = rows.astype(np.float16).view(np.uint16)
    lut = np.array([L.ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
    img = oracle_image(dict(a, rows=lut[h]))
    full = IvfIndex(dim, nlists)
    full.set_centroids(a["centroids"])
    full.load(a["list_len"], a["rows"], a["tids"])
    twin = full.to_f16(reference_encoder)
    q = _queries(a, 40, seed=78)
    for strategy in (3, 1, 2):
        et, ed, ec, _ = oracle_search_batch(img, q, strategy, 6, 10)
        for mode in (0, 3, 5):                          # 3: screened — the twin without subnormals decodes with the plain conversion
            scan_mode(mode)
            t, d, c = twin.search(q, strategy, 6, 10)
            assert_same_results(t, d, c, et, ed, ec)
    scan_mode(0)
    # shards of the fp16 twin + merge == the twin
    world, k, nprobe = 2, 10, 6
    cap = 3 * k
    dq = torch.from_numpy(q).cuda()
    cand = torch.zeros((world, len(q), cap, 2), dtype=torch.int64, device="cuda")
    ncand = torch.zeros((world, len(q)), dtype=torch.int32, device="cuda")
    total = torch.zeros((world, len(q)), dtype=torch.int64, device="cuda")
    keep = []
    for w in range(world):
        sh = twin.shard((np.arange(nlists) % world == w).astype(np.uint8))
        sh.search_partial_device(dq, cand[w], ncand[w], total[w], 3, nprobe, k)
        keep.append(sh)
    ot = torch.zeros((len(q), k), dtype=torch.int64, device="cuda")
    od = torch.zeros((len(q), k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(len(q), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ndbhip_merge_topk_device(cand.data_ptr(), ncand.data_ptr(), total[0].data_ptr(),
                                                   world, len(q), k, cap, ot.data_ptr(), od.data_ptr(),
                                                   oc.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    et, ed, ec, _ = oracle_search_batch(img, q, 3, nprobe, k)
    assert np.array_equal(oc.cpu().numpy(), ec)
    assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
    assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))


def test_bulkdelete_drops_rows_and_keeps_list_order(scan_mode):
    """ndbhip_ivf_delete = ambulkdelete's effect on later scans (dead line pointers are skipped,
    ivf_am.c:1816-1822): results equal the oracle's over the arrays without the deleted entries; a whole list
    can vanish; unknown TIDs are ignored; appends keep working afterwards."""
    from neurondb_amd import IvfIndex
    a = make_ivf_arrays(6000, 64, 24, seed=91, dup_frac=0.1)
    n = len(a["rows"])
    rng = np.random.default_rng(92)
    off = np.zeros(25, np.int64)
    off[1:] = np.cumsum(a["list_len"])
    dead = np.zeros(n, bool)
    dead[rng.choice(n, 900, replace=False)] = True
    dead[off[5]:off[6]] = True                              # list 5 loses every entry
    ix = IvfIndex(64, 24)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    ghost = a["tids"][:3].copy()
    ghost["bi_lo"] = 65000                                  # TIDs the index does not hold
    assert ix.delete(ghost) == 0
    victims = np.concatenate([a["tids"][dead], ghost])
    assert ix.delete(victims[rng.permutation(len(victims))]) == int(dead.sum())
    assert ix.delete(a["tids"][dead][:10]) == 0             # already gone
    keep = ~dead
    new_len = np.array([keep[off[l]:off[l + 1]].sum() for l in range(24)], np.int64)
    b = dict(a, rows=a["rows"][keep], tids=a["tids"][keep], list_len=new_len)
    cent, ll, rows, tids = ix.export()
    assert np.array_equal(ll, new_len) and np.array_equal(rows, b["rows"]) and np.array_equal(tids, b["tids"])
    img = oracle_image(b)
    q = _queries(a, 40, seed=93)
    for mode in (1, 2, 3, 5):
        scan_mode(mode)
        for strategy in (1, 2, 3):
            t, d, c = ix.search(q, strategy, 6, 10)
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, 6, 10)
            assert_same_results(t, d, c, et, ed, ec)
    # aminsert after vacuum: the entry lands at the tail of its list
    from oracle import ndbo
    newtid = ghost[:1]
    vnew = (a["rows"][0] + 37.0).astype(np.float32)
    ix.append(5, vnew, newtid)
    t, d, c = ix.search(vnew[None, :], 1, 24, 1)
    assert c[0] == 1 and d[0, 0] == 0.0 and ndbo.tids_to_u64(t[0, :1])[0] == ndbo.tids_to_u64(newtid)[0]
    assert ix.export()[1][5] == 1


def _merged_partial(shards, dq, strategy, nprobe, k, max_candidates=0, probes=None):
    """per-shard partial records -> device merge (what dist.sharded_search does around the all-gather)"""
    import torch
    from neurondb_amd import _lib
    from oracle import ndbo
    world, nq, cap = len(shards), dq.shape[0], 3 * k
    cand = torch.zeros((world, nq, cap, 2), dtype=torch.int64, device="cuda")
    ncand = torch.zeros((world, nq), dtype=torch.int32, device="cuda")
    total = torch.zeros((world, nq), dtype=torch.int64, device="cuda")
    for w, sh in enumerate(shards):
        if probes is None:
            sh.search_partial_device(dq, cand[w], ncand[w], total[w], strategy, nprobe, k, max_candidates)
        else:
            sh.search_partial_probes_device(dq, probes, cand[w], ncand[w], total[w], strategy, nprobe, k,
                                            max_candidates)
    ot = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    od = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(nq, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ndbhip_merge_topk_device(cand.data_ptr(), ncand.data_ptr(), total[0].data_ptr(),
                                                   world, nq, k, cap, ot.data_ptr(), od.data_ptr(), oc.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    tot = total.cpu().numpy()
    assert np.array_equal(tot, np.broadcast_to(tot[0], tot.shape))     # every rank knows the global count
    return ndbo.tids_from_device_u64(ot.cpu().numpy()), od.cpu().numpy(), oc.cpu().numpy()


def _same_u64(t, d, c, et, ed, ec):
    assert np.array_equal(c, ec), (c, ec)
    for i in range(len(ec)):
        n = ec[i]
        assert np.array_equal(t[i, :n], et[i, :n]), (i, t[i, :n], et[i, :n], d[i, :n], ed[i, :n])
        assert np.array_equal(d[i, :n].view(np.uint32), ed[i, :n].view(np.uint32)), (i, d[i, :n], ed[i, :n])


@pytest.mark.parametrize("world,split_frac,align", [(4, 0.0, 16), (3, 0.0, 64), (8, 0.25, 64)])
def test_slice_shards_merge_to_the_unsharded_result(world, split_frac, align, scan_mode):
    """Lists cut into per-rank slices (partition_slices / ndbhip_ivf_shard_slices): the merged result is the
    oracle's for all strategies, with and without the candidate cap, for batches on the grouped and on the
    per-query scan, for fp16 rows, and with the probes chosen elsewhere."""
    import torch
    from neurondb_amd.dist import partition_slices
    nl = 16
    a = make_ivf_arrays(7000, 64, nl, seed=131, dup_frac=0.1, empty_lists=(3,))
    img = oracle_image(a)
    full = _index(a)
    pc = np.ones(nl)
    pc[np.argmax(a["list_len"])] = 50
    lo, ln, tail = partition_slices(a["list_len"], world, pc, split_frac=split_frac, align=align)
    assert (ln.sum(0) == a["list_len"]).all() and (tail.sum(0) == 1).all()
    if split_frac == 0.0:
        assert ((ln > 0).sum(0) > 1).any()                  # some list really is split
    shards = [full.shard_slices(lo[w], ln[w], tail[w]) for w in range(world)]
    assert sum(s.nrows for s in shards) == full.nrows
    k, nprobe = 10, 6
    for nq in (100, 3):
        q = _queries(a, nq, seed=132 + nq)
        dq = torch.from_numpy(q).cuda()
        for strategy in (1, 2, 3):
            for cap in (0, 700):
                for mode in (1, 2, 3, 5):
                    scan_mode(mode)
                    t, d, c = _merged_partial(shards, dq, strategy, nprobe, k, cap)
                    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
                    _same_u64(t, d, c, et, ed, ec)
        probes = torch.from_numpy(np.stack([img.select_clusters(qq, nprobe) for qq in q]).astype(np.int32)).cuda()
        t, d, c = _merged_partial(shards, dq, 1, nprobe, k, 0, probes)
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
        _same_u64(t, d, c, et, ed, ec)
    # fp16 twins of the slices
    twin = full.to_f16(False)
    from oracle import ndbo
    lut = np.array([ndbo.lib().ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
    img16 = oracle_image(dict(a, rows=lut[a["rows"].astype(np.float16).view(np.uint16)]))
    sh16 = [twin.shard_slices(lo[w], ln[w], tail[w]) for w in range(world)]
    q = _queries(a, 40, seed=139)
    dq = torch.from_numpy(q).cuda()
    for strategy in (1, 3):
        t, d, c = _merged_partial(sh16, dq, strategy, nprobe, k)
        et, ed, ec, _ = oracle_search_batch(img16, q, strategy, nprobe, k)
        _same_u64(t, d, c, et, ed, ec)
    # a slice of a slice (re-sharding a shard) and the bounds check
    sub = shards[0].shard_slices(lo[0], np.minimum(ln[0], 5))
    assert sub.nrows == int(np.minimum(ln[0], 5).sum())
    with pytest.raises(Exception):
        shards[0].shard_slices(lo[0], ln[0] + 1)


def test_slice_shards_take_appends_on_the_tail_rank():
    """aminsert after slicing: every rank is told of the append (list lengths are global), the rank holding the
    list's tail stores the row; merged results equal the oracle's over the grown lists."""
    import torch
    from neurondb_amd.dist import partition_slices
    from oracle import ndbo
    nl, world = 12, 3
    a = make_ivf_arrays(4000, 64, nl, seed=141, empty_lists=(2,))
    full = _index(a)
    lo, ln, tail = partition_slices(a["list_len"], world, None, split_frac=0.0, align=32)
    shards = [full.shard_slices(lo[w], ln[w], tail[w]) for w in range(world)]
    rng = np.random.default_rng(142)
    off = np.zeros(nl + 1, np.int64)
    off[1:] = np.cumsum(a["list_len"])
    lists = [list(range(off[l], off[l + 1])) for l in range(nl)]
    rows, tids = [a["rows"]], [a["tids"]]
    nbase = len(a["rows"])
    for i, l in enumerate([0, 2, 2, 7, 0, 11, 5]):                     # list 2 starts empty
        v = rng.standard_normal(64).astype(np.float32)
        t = ndbo.tids_from_rows(np.array([900000 + i]))
        for s in shards:
            s.append(l, v, t)
        lists[l].append(nbase + i)
        rows.append(v[None])
        tids.append(t)
    rows, tids = np.concatenate(rows), np.concatenate(tids)
    order = np.concatenate([np.asarray(x, np.int64) for x in lists])
    grown = dict(a, rows=rows[order], tids=tids[order], list_len=np.array([len(x) for x in lists], np.int64))
    img = oracle_image(grown)
    q = _queries(a, 50, seed=143)
    dq = torch.from_numpy(q).cuda()
    t, d, c = _merged_partial(shards, dq, 1, 5, 10)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 5, 10)
    _same_u64(t, d, c, et, ed, ec)
    assert sum(s.nrows for s in shards) == nbase + 7


@pytest.mark.parametrize("case", ["cancellation", "ties", "spread_norms", "wide", "tiny"])
def test_screened_scan_is_exact_where_its_bound_is_weakest(case, scan_mode):
    """The screened L2 scan (mode 3: fused dot + norms as a lower bound, the reference's arithmetic for the
    survivors) against the oracle on data chosen to stress the bound: rows far from the origin and close to each
    other (the expansion |q|^2 + |x|^2 - 2 q.x cancels almost completely, so nearly everything survives and the
    per-query survivor list overflows into the in-place path), exact ties and duplicates, norms spread over six
    orders of magnitude, a wide dimension, and tiny values."""
    rng = np.random.default_rng(hash(case) % 1000)
    dim, n, nlists, nq = 64, 6000, 10, 70
    if case == "cancellation":
        center = rng.standard_normal(dim).astype(np.float32) * 300.0
        base = (center + rng.standard_normal((n, dim)).astype(np.float32) * 1e-2).astype(np.float32)
        q = (center + rng.standard_normal((nq, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    elif case == "ties":
        base = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
        base[n // 2:] = base[: n - n // 2]
        q = rng.integers(-2, 3, size=(nq, dim)).astype(np.float32)
        q[:5] = base[:5]
    elif case == "spread_norms":
        base = (rng.standard_normal((n, dim)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
        base[7] = 0.0
        q = (rng.standard_normal((nq, dim)) * 10.0 ** rng.uniform(-3, 3, (nq, 1))).astype(np.float32)
        q[3] = 0.0
    elif case == "wide":
        dim, n, nlists = 1536, 3000, 6
        base = rng.standard_normal((n, dim)).astype(np.float32)
        q = (base[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    else:
        base = (rng.standard_normal((n, dim)) * 1e-18).astype(np.float32)
        q = (rng.standard_normal((nq, dim)) * 1e-18).astype(np.float32)
    cent = base[rng.choice(n, nlists, replace=False)].copy()
    d2 = ((base[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1) if dim <= 64 else \
        (base.astype(np.float64) ** 2).sum(1)[:, None] + (cent.astype(np.float64) ** 2).sum(1)[None] - \
        2.0 * base.astype(np.float64) @ cent.astype(np.float64).T
    asg = d2.argmin(1)
    order = np.argsort(asg, kind="stable")
    from oracle import ndbo
    a = dict(centroids=cent, list_len=np.bincount(asg, minlength=nlists).astype(np.int64),
             rows=np.ascontiguousarray(base[order]), tids=ndbo.tids_from_rows(order))
    img = oracle_image(a)
    ix = _index(a)
    for k, nprobe, cap in ((10, 4, 0), (100, nlists, 0), (10, nlists, 300)):
        for strategy in (1, 3, 2):                              # inner product and cosine are screened too
            if strategy != 1 and case == "tiny":
                continue                                          # products of 1e-18 values underflow to NaN-free zeros: L2 only
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
            for mode in (3, 2, 5):
                scan_mode(mode)
                t, d, c = ix.search(q, strategy, nprobe, k, cap)
                assert_same_results(t, d, c, et, ed, ec)
    from neurondb_amd import _lib
    scan_mode(3)
    _lib.check(_lib.lib().ndbhip_stats_reset())
    ix.search(q, 1, nlists, 10)
    st = _lib.stats()
    assert 0 < st["rows_rescored"] <= nq * n                  # the second pass ran
    if case == "cancellation":
        assert st["rows_rescored"] > nq * 256                 # ... and overflowed the per-query list


def test_randomised_parity_campaign_all_scan_modes():
    """A few seconds of tools/fuzz_scan.py (random shapes, scales, k, nprobe, caps, strategies; scan modes screened /
    grouped / per-query / auto) — the long runs are in profiles/r01l_fuzz_scan.txt."""
    import time
    from neurondb_amd import IvfIndex, _lib
    from tools.fuzz_scan import one_case, reset_options
    rng = np.random.default_rng(12345)
    t0, n = time.time(), 0
    try:
        while time.time() - t0 < 6.0:
            one_case(rng, _lib.lib(), IvfIndex, _lib.check)
            n += 1
    finally:
        reset_options(_lib.lib(), _lib.check)        # (one_case draws a dozen options: whatever the last case drew must not stay)
    assert n >= 20


@pytest.mark.parametrize("dim,nq", [(64, 300), (100, 37), (33, 9)])
def test_queries_gathered_from_mapped_host_memory_give_the_same_answers(dim, nq):
    """ndbhip_ivf_search_mapped (what the device-owner service serves its ring with): the queries lie scattered in pinned
    host memory — other bytes between them, not in order, some of them not 16-byte aligned — and a kernel gathers them;
    the answers are ndbhip_ivf_search's, hence the oracle's."""
    import torch
    from neurondb_amd import IvfIndex, _lib
    _lib.ensure_init()
    n, nlists = 4000, 12
    a = make_ivf_arrays(n, dim, nlists, seed=dim + 3, dup_frac=0.05)
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    rng = np.random.default_rng(dim)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[: nq // 2] = a["rows"][rng.integers(0, n, nq // 2)]
    stride = dim * 4 + 4 * int(rng.integers(1, 40))            # a slot: the query and some bytes of something else
    ring = torch.zeros(nq * stride + 64, dtype=torch.uint8).pin_memory()
    order = rng.permutation(nq)                                 # query i lives in slot order[i]
    offs = (order * stride + 4).astype(np.int64)                # (+ 4: every other slot start is not 16-byte aligned)
    host = ring.numpy()
    for i in range(nq):
        host[offs[i]: offs[i] + dim * 4] = q[i].view(np.uint8)
    for strategy, nprobe, k in ((1, 5, 10), (3, nlists, 7), (2, 3, 1)):
        et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k, 0)
        t, d, c = ix.search_mapped(ring.data_ptr(), offs, strategy, nprobe, k)
        assert_same_results(t, d, c, et, ed, ec)
    ix.close()
