"""Parity of the HIP IVF build (k-means, insert-time assignment, list packing)
against the CPU oracle: centroids bit-identical, every row in the same list at
the same position."""
import ctypes as C

import numpy as np
import pytest

from oracle import ndbo

pytestmark = pytest.mark.gpu


def _torch(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n,dim,k", [(300, 4, 5), (2000, 28, 10), (1500, 6, 7), (3000, 100, 16), (900, 768, 8)])
def test_kmeans_matches_oracle(n, dim, k):
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    rng = np.random.default_rng(n + dim)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    data[5] = data[3]                       # duplicate sample
    cent, asg, cnt, iters, cost = ndbo.kmeans(data, k, max_iter=50)
    d = _torch(data)
    dc = torch.zeros((k, dim), dtype=torch.float32, device="cuda")
    da = torch.zeros(n, dtype=torch.int32, device="cuda")
    dn = torch.zeros(k, dtype=torch.int32, device="cuda")
    it = C.c_int(0)
    cs = C.c_float(0)
    _lib.check(_lib.lib().ndbhip_kmeans_device(d.data_ptr(), n, dim, k, 50, np.float32(0.001), dc.data_ptr(),
                                               da.data_ptr(), dn.data_ptr(), C.byref(it), C.byref(cs)))
    assert it.value == iters
    assert np.array_equal(da.cpu().numpy(), asg)
    assert np.array_equal(dn.cpu().numpy(), cnt)
    assert np.array_equal(dc.cpu().numpy().view(np.uint32), cent.view(np.uint32))
    assert np.float32(cs.value).tobytes() == np.float32(cost).tobytes()


def test_kmeans_empty_cluster_and_k_gt_distinct():
    """identical samples => every point goes to centroid 0, others stay all-zero (ivf_am.c:2189-2212)."""
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    data = np.ones((50, 8), np.float32)
    cent, asg, cnt, iters, cost = ndbo.kmeans(data, 4)
    d = _torch(data)
    dc = torch.zeros((4, 8), dtype=torch.float32, device="cuda")
    da = torch.zeros(50, dtype=torch.int32, device="cuda")
    dn = torch.zeros(4, dtype=torch.int32, device="cuda")
    it = C.c_int(0)
    _lib.check(_lib.lib().ndbhip_kmeans_device(d.data_ptr(), 50, 8, 4, 50, np.float32(0.001), dc.data_ptr(),
                                               da.data_ptr(), dn.data_ptr(), C.byref(it), None))
    assert it.value == iters and np.array_equal(dn.cpu().numpy(), cnt)
    assert np.array_equal(dc.cpu().numpy().view(np.uint32), cent.view(np.uint32))


@pytest.mark.parametrize("n,dim,nlists", [(10000, 128, 100), (3000, 28, 10), (1000, 6, 4), (5000, 768, 24)])
def test_build_matches_oracle_build(n, dim, nlists):
    """BASELINE configs[0] shape (10k x 128, lists=100) and friends: whole build."""
    from neurondb_amd import IvfIndex
    rng = np.random.default_rng(dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    img, asg, iters = ndbo.build_ivf_image(base, nlists, max_iter=50)
    ix = IvfIndex(dim, nlists)
    it = ix.build(base, ndbo.tids_from_rows(np.arange(n)))
    cent, ll, rows, tids = ix.export()
    assert it == iters
    assert np.array_equal(cent.view(np.uint32), img.centroids.view(np.uint32))
    assert np.array_equal(ll, np.diff(img.list_off))
    assert np.array_equal(ndbo.tids_to_u64(tids), ndbo.tids_to_u64(img.tids))
    assert np.array_equal(rows.view(np.uint32), img.vecs.view(np.uint32))
    # and the built index answers queries like the oracle's
    q = rng.standard_normal((8, dim)).astype(np.float32)
    from tests.util import assert_same_results, oracle_search_batch
    t, d, c = ix.search(q, 1, min(10, nlists), 10)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, min(10, nlists), 10)
    assert_same_results(t, d, c, et, ed, ec)


def test_assign_device_matches_insert_rule():
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    rng = np.random.default_rng(1)
    cent = rng.standard_normal((130, 64)).astype(np.float32)
    cent[7] = cent[3]                                   # duplicate centroid: first wins
    rows = rng.standard_normal((700, 64)).astype(np.float32)
    rows[:130] = cent                                   # exact hits
    exp = ndbo.ivf_assign_all(cent, rows)
    out = torch.zeros(700, dtype=torch.int32, device="cuda")
    dc, dr = _torch(cent), _torch(rows)
    _lib.check(_lib.lib().ndbhip_ivf_assign_device(dc.data_ptr(), 130, 64, dr.data_ptr(), 700, out.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    assert np.array_equal(out.cpu().numpy(), exp)


@pytest.mark.parametrize("n,dim,k,kind", [
    (6000, 64, 130, "gauss"),          # few rows a tile, ragged last tiles
    (5000, 768, 300, "clustered"),     # tight clusters: the bound leaves one centroid for most rows
    (4500, 100, 40, "dups"),           # a centroid repeated 12 times: more candidates than record slots
    (4200, 64, 70, "special"),         # NaN / inf / huge rows and a NaN centroid
    (4100, 30, 5, "gauss"),            # dim not a multiple of 4, fewer centroids than a tile
])
def test_screened_assignment_matches_insert_rule(n, dim, k, kind):
    """ndbhip_ivf_assign_device with >= 4096 rows goes through the fp16 matrix-core sweeps (assign_rows_s16); the
    lists must be the insert rule's (oracle ivf_assign_all), and the same as with build_screen16=0."""
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    rng = np.random.default_rng(n + dim)
    cent = rng.standard_normal((k, dim)).astype(np.float32)
    if kind == "clustered":
        rows = (cent[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
        cent[11] = cent[10] + np.float32(1e-4)        # two centroids the bound cannot separate
    else:
        rows = rng.standard_normal((n, dim)).astype(np.float32)
    if kind == "dups":
        cent[20:32] = cent[5]
        rows[:500] = cent[5] + 0.01 * rng.standard_normal((500, dim)).astype(np.float32)
    if kind == "special":
        rows[0, 3] = np.nan
        rows[1, 0] = np.inf
        rows[2] = 3e19                                   # the squared norm overflows fp32
        rows[3] = 0
        rows[4] = 1e-30
        cent[9, 1] = np.nan
        cent[12] = 0
    rows[100:100 + k] = cent                             # exact hits
    with np.errstate(all="ignore"):
        exp = ndbo.ivf_assign_all(cent, rows)
    dc, dr = _torch(cent), _torch(rows)
    got = {}
    try:
        for opt in (1, 0):
            _lib.check(_lib.lib().ndbhip_set_option(b"build_screen16", opt))
            out = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            _lib.check(_lib.lib().ndbhip_ivf_assign_device(dc.data_ptr(), k, dim, dr.data_ptr(), n, out.data_ptr()))
            _lib.check(_lib.lib().ndbhip_synchronize())
            got[opt] = out.cpu().numpy()
    finally:
        _lib.check(_lib.lib().ndbhip_set_option(b"build_screen16", 1))
    assert np.array_equal(got[0], exp)
    assert np.array_equal(got[1], exp)


def test_append_keeps_lists_in_insertion_order_and_matches_oracle():
    """aminsert path: 70 % of the rows loaded, 30 % appended one by one to the list ivfinsert
    would choose; the result must equal the oracle image holding all rows."""
    from neurondb_amd import IvfIndex
    from tests.util import assert_same_results, oracle_search_batch
    rng = np.random.default_rng(5)
    n, dim, nlists = 2000, 32, 9
    base = rng.standard_normal((n, dim)).astype(np.float32)
    cent = base[rng.choice(n, nlists, replace=False)].copy()
    asg = ndbo.ivf_assign_all(cent, base)
    n0 = 1400
    tids = ndbo.tids_from_rows(np.arange(n))
    order0 = np.argsort(asg[:n0], kind="stable")
    ll0 = np.bincount(asg[:n0], minlength=nlists).astype(np.int64)
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(cent)
    ix.load(ll0, base[:n0][order0], tids[:n0][order0])
    for r in range(n0, n):
        ix.append(asg[r], base[r], tids[r])
    assert ix.nrows == n
    order = np.argsort(asg, kind="stable")
    off = np.zeros(nlists + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(asg, minlength=nlists))
    img = ndbo.IvfImage(cent, off, base[order], tids[order])
    q = rng.standard_normal((10, dim)).astype(np.float32)
    t, d, c = ix.search(q, 1, 4, 10)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 4, 10)
    assert_same_results(t, d, c, et, ed, ec)
    _, ll, rows, tt = ix.export()
    assert np.array_equal(ll, np.diff(off))
    assert np.array_equal(rows.view(np.uint32), img.vecs.view(np.uint32))
    assert np.array_equal(ndbo.tids_to_u64(tt), ndbo.tids_to_u64(img.tids))


def test_host_pointer_build_is_the_device_build():
    """ndbhip_ivf_build (rows and heapPtrs in host memory, staged by the library) == the oracle's build:
    what an ambuild written in C calls, with no device allocation of its own."""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    rng = np.random.default_rng(3)
    n, dim, nlists = 4000, 48, 16
    base = rng.standard_normal((n, dim)).astype(np.float32)
    img, asg, iters = ndbo.build_ivf_image(base, nlists, max_iter=50)
    ix = IvfIndex(dim, nlists)
    t6 = np.ascontiguousarray(ndbo.tids_from_rows(np.arange(n))).view(np.uint8).reshape(-1, 6)
    it = C.c_int()
    _lib.check(_lib.lib().ndbhip_ivf_build(ix._h, base.ctypes.data, t6.ctypes.data, n, 50, C.byref(it)))
    ix.ncent = nlists
    cent, ll, rows, tids = ix.export()
    assert it.value == iters
    assert np.array_equal(cent.view(np.uint32), img.centroids.view(np.uint32))
    assert np.array_equal(ll, np.diff(img.list_off))
    assert np.array_equal(ndbo.tids_to_u64(tids), ndbo.tids_to_u64(img.tids))
    assert np.array_equal(rows.view(np.uint32), img.vecs.view(np.uint32))


def test_build_from_host_memory_equals_build_from_device_memory_and_can_leave_the_index_prepared():
    """ndbhip_ivf_build uploads the table in heap order while the k-means and the assignment of the slabs that have
    arrived already run (csrc/ndbhip_build.h); 200k x 768 is several upload pieces and two assignment slabs.  The
    index must be the one ndbhip_ivf_build_device makes from the same rows; with option build_prepare the sublists /
    planes are made inside the build, not by the first batch of queries."""
    import torch
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import Stats, check, lib
    from bench import make_data, pack_tids
    _lib.ensure_init()
    n, dim, nlists = 200_000, 768, 64
    dev = torch.device("cuda", 0)
    base = make_data(n, dim, "clustered", 64, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    tids = pack_tids(torch.arange(n, device=dev))
    a = IvfIndex(dim, nlists)
    ita = a.build_device(base, tids, 50)
    check(lib().ndbhip_synchronize())
    ca, la, ra, ta = a.export()
    host = base.cpu().numpy()
    st = Stats()
    try:
        check(lib().ndbhip_set_option(b"build_prepare", 1))
        check(lib().ndbhip_stats_reset())
        b = IvfIndex(dim, nlists)
        itb = b.build(host, tids.cpu().numpy(), 50)
        check(lib().ndbhip_stats_get(C.byref(st)))
        prepared_in_build = st.prepares
    finally:
        check(lib().ndbhip_set_option(b"build_prepare", 0))
    cb, lb, rb, tb = b.export()
    assert ita == itb
    assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32))
    assert np.array_equal(la, lb)
    assert np.array_equal(ndbo.tids_to_u64(ta), ndbo.tids_to_u64(tb))
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
    assert prepared_in_build == 1
    q = make_data(256, dim, "clustered", 64, 0.1, 0x5EED0002, 0x5EEDC0DE, dev).cpu().numpy()
    t1, d1, c1 = a.search(q, 1, 8, 10)
    t2, d2, c2 = b.search(q, 1, 8, 10)
    check(lib().ndbhip_stats_get(C.byref(st)))
    assert st.prepares == 2                     # a's first batch prepared a; b was ready
    assert np.array_equal(c1, c2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
    assert np.array_equal(ndbo.tids_to_u64(t1.reshape(-1)), ndbo.tids_to_u64(t2.reshape(-1)))
