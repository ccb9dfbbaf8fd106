"""BASELINE.json's full sizes on the device, checked through properties that do not need the CPU oracle to
redo the whole job (it takes 42 ms per query at this size): results are sorted, every returned distance is the
oracle's scalar distance of the returned row, a query's result does not depend on the batch it travels in (one
query = per-query kernel, 8 = grouped scan with look-before-take, 4096 = grouped scan) nor on how the lists are
cut over ranks, repeated calls agree bit for bit, and a sample of queries is replayed in full by the oracle.

  C2: 1M x 768 fp32, IVFFlat lists = 1024, probes = 32, k = 10, L2   (configs[1], the bench workload)
  C3: 1M x 768 unit-norm rows, HNSW m = 16, ef_construction = 200, ef_search = 64, k = 10, cosine
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_data, pack_tids, unpack_tids
from oracle import ndbo

pytestmark = pytest.mark.gpu

N, DIM, LISTS, PROBES, K = 1_000_000, 768, 1024, 32, 10


@pytest.fixture(scope="module")
def c2():
    from neurondb_amd import IvfIndex, _lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    base = make_data(N, DIM, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(4096, DIM, "clustered", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(DIM, LISTS)
    ix.build_device(base, pack_tids(torch.arange(N, device=dev)), 50)
    base_h = base.cpu().numpy()
    del base
    yield ix, q, base_h
    ix.close()


def _search(ix, q, strategy=1):
    from neurondb_amd import _lib
    nq = q.shape[0]
    ot = torch.zeros((nq, K), dtype=torch.int64, device=q.device)
    od = torch.zeros((nq, K), dtype=torch.float32, device=q.device)
    oc = torch.zeros(nq, dtype=torch.int32, device=q.device)
    ix.search_device(q.contiguous(), ot, od, oc, strategy, PROBES, K, 0)
    _lib.check(_lib.lib().ndbhip_synchronize())
    return unpack_tids(ot).cpu().numpy(), od.cpu().numpy(), oc.cpu().numpy()


def test_c2_properties_at_full_size(c2):
    ix, q, base_h = c2
    rows, dist, cnt = _search(ix, q)
    assert (cnt == K).all()
    # sorted: the reference's selection sort emits non-decreasing distances
    assert (np.diff(dist, axis=1) >= 0).all()
    # a query never returns a row twice
    assert all(len(set(r.tolist())) == K for r in rows[:512])
    # every distance is ivfComputeDistance(query, that row), bit for bit (oracle scalar recipe)
    L = ndbo.lib()
    qh = q.cpu().numpy()
    for i in range(0, 4096, 37):
        exp = np.array([L.ndbo_ivf_distance(qh[i], base_h[r], DIM, 1) for r in rows[i]], np.float32)
        assert np.array_equal(exp.view(np.uint32), dist[i].view(np.uint32)), i
    # idempotence
    rows2, dist2, cnt2 = _search(ix, q)
    assert np.array_equal(rows, rows2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32))
    # the batch a query travels in does not matter: alone (per-query kernel), in 3 (per-query), in 8 and 100
    # (grouped scan, small-batch queue policy), in 4096 (grouped scan)
    for lo, n in ((0, 1), (5, 1), (40, 3), (64, 8), (1000, 100), (4000, 96)):
        r, d, c = _search(ix, q[lo:lo + n])
        assert np.array_equal(r, rows[lo:lo + n]) and np.array_equal(d.view(np.uint32), dist[lo:lo + n].view(np.uint32))
    # cosine and inner product: sorted, and equal for single queries and the batch
    for strategy in (2, 3):
        r_all, d_all, _ = _search(ix, q[:512], strategy)
        assert (np.diff(d_all, axis=1) >= 0).all()
        r1, d1, _ = _search(ix, q[7:8], strategy)
        assert np.array_equal(r1[0], r_all[7]) and np.array_equal(d1[0].view(np.uint32), d_all[7].view(np.uint32))


def test_c2_oracle_replay_of_a_sample_at_full_size(c2):
    """16 queries replayed in full by the CPU oracle on the exported index (ids, ranks, float4 bits)."""
    from concurrent.futures import ThreadPoolExecutor
    ix, q, _ = c2
    rows, dist, cnt = _search(ix, q[:16])
    cent, list_len, rows_h, tid_h = ix.export(rows=True)
    off = np.zeros(LISTS + 1, np.int64)
    off[1:] = np.cumsum(list_len)
    img = ndbo.IvfImage(cent, off, rows_h, np.ascontiguousarray(tid_h).view(ndbo.TID_DTYPE).reshape(-1))
    qh = q[:16].cpu().numpy()
    ndbo.lib()
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda i: img.search(qh[i], 1, PROBES, K, 0), range(16)))
    for i, (et, ed, _) in enumerate(res):
        er = (((et["bi_hi"].astype(np.int64) << 16) | et["bi_lo"]) * 64 + et["posid"] - 1)
        assert np.array_equal(er, rows[i]) and np.array_equal(ed.view(np.uint32), dist[i].view(np.uint32)), i


def test_c2_slices_over_8_ranks_merge_to_the_unsharded_result(c2):
    """The 8-GPU partition of the bench (partition_slices on calibration counts), one shard after the other on
    this GPU: partial records -> merge == the single-device result for the whole 4096-query batch."""
    from neurondb_amd import _lib
    from neurondb_amd.dist import ShardedSearchBuffers, partition_slices
    ix, q, _ = c2
    rows, dist, cnt = _search(ix, q)
    world = 8
    dev = q.device
    _, ll, _, _ = ix.export(rows=False)
    probes = torch.zeros((4096, PROBES), dtype=torch.int32, device=dev)
    ix.select_clusters_device(q, probes, PROBES)
    _lib.check(_lib.lib().ndbhip_synchronize())
    pc = probes.cpu().numpy()
    lo, ln, tail = partition_slices(ll, world, np.bincount(pc.ravel(), minlength=LISTS))
    assert ((ln > 0).sum(0) > 1).any()                        # the 28 k-row list is cut
    buf = ShardedSearchBuffers(4096, K, world, dev, nprobe=PROBES)
    for w in range(world):
        sh = ix.shard_slices(lo[w], ln[w], tail[w])
        sh.search_partial_probes_device(q, probes, buf.cand, buf.ncand, buf.total, 1, PROBES, K, 0)
        _lib.check(_lib.lib().ndbhip_synchronize())
        buf.cand_all[w].copy_(buf.cand)
        buf.ncand_all[w].copy_(buf.ncand)
        sh.close()
    _lib.check(_lib.lib().ndbhip_merge_topk_device(buf.cand_all.data_ptr(), buf.ncand_all.data_ptr(),
                                                   buf.total.data_ptr(), world, 4096, K, buf.cap,
                                                   buf.out_tids.data_ptr(), buf.out_dist.data_ptr(),
                                                   buf.out_count.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    assert np.array_equal(unpack_tids(buf.out_tids).cpu().numpy(), rows)
    assert np.array_equal(buf.out_dist.cpu().numpy().view(np.uint32), dist.view(np.uint32))
    assert np.array_equal(buf.out_count.cpu().numpy(), cnt)


def test_c3_hnsw_at_full_size():
    """1M-node graph built on the device: both search kernels and the 8f-2 search agree with themselves across
    batch sizes, every distance is the oracle's scalar recipe on the returned node, the build is deterministic
    (same neighbour-array digest twice), and a sample is replayed by the oracle on the exported graph."""
    import ctypes as C
    import hashlib
    from neurondb_amd import HnswIndex, _lib
    lib, check = _lib.lib(), _lib.check
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    n, m, efc, ef = N, 16, 200, 64
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0003)
    base = torch.randn((n, DIM), generator=g, device=dev)
    base = base / base.norm(dim=1, keepdim=True)
    q = torch.randn((2048, DIM), generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)
    tids = pack_tids(torch.arange(n, device=dev))

    def build():
        ix = HnswIndex(DIM, m)
        check(lib.ndbhip_hnsw_build_device(ix._h, C.c_void_p(base.data_ptr()), C.c_void_p(tids.data_ptr()), n,
                                           levels.ctypes.data, efc))
        check(lib.ndbhip_synchronize())
        return ix

    def run(ix, qs, strategy=2, layer=False):
        nq = qs.shape[0]
        ob = torch.zeros((nq, K), dtype=torch.int32, device=dev)
        od = torch.zeros((nq, K), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        sc = torch.zeros(nq, dtype=torch.int64, device=dev)
        fn = lib.ndbhip_hnsw_search_layer_device if layer else lib.ndbhip_hnsw_search_device
        check(fn(ix._h, C.c_void_p(qs.contiguous().data_ptr()), nq, strategy, ef, K, C.c_void_p(ob.data_ptr()),
                 C.c_void_p(od.data_ptr()), C.c_void_p(oc.data_ptr()), None, C.c_void_p(sc.data_ptr())))
        check(lib.ndbhip_synchronize())
        return ob.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy(), sc.cpu().numpy()

    ix = build()
    e = ix.export()
    digest = hashlib.sha256(e["nbrs"].tobytes() + e["ncount"].tobytes()).hexdigest()
    ob, od, oc, sc = run(ix, q)
    # the reference's level-0 walk can end with fewer than k candidates (Q10); what it returns is sorted
    assert (oc >= 1).all() and (oc <= K).all() and (oc == K).mean() > 0.9
    assert all((np.diff(od[i, :oc[i]]) >= 0).all() for i in range(len(oc)))
    # both kernels, and any batch size, give the same bits
    try:
        check(lib.ndbhip_hnsw_set_search_mode(1))
        ob1, od1, oc1, sc1 = run(ix, q[:256])
    finally:
        check(lib.ndbhip_hnsw_set_search_mode(0))
    assert np.array_equal(oc1, oc[:256]) and np.array_equal(sc1, sc[:256])
    assert np.array_equal(ob1, ob[:256]) and np.array_equal(od1.view(np.uint32), od[:256].view(np.uint32))
    b1, d1, _, _ = run(ix, q[9:10])
    assert np.array_equal(b1[0], ob[9]) and np.array_equal(d1[0].view(np.uint32), od[9].view(np.uint32))
    # distances are hnswComputeDistance (cosine) of the returned nodes
    L = ndbo.lib()
    bh, qh = base.cpu().numpy(), q.cpu().numpy()
    err = C.c_int(0)
    for i in range(0, 2048, 101):
        exp = np.array([L.ndbo_hnsw_distance(qh[i], bh[b - 1], DIM, 2, C.byref(err)) for b in ob[i, :oc[i]]], np.float32)
        assert np.array_equal(exp.view(np.uint32), od[i, :oc[i]].view(np.uint32)), i
    # hnsw_search_layer (8f-2): same bits alone and in the batch; compute_l2_distance of the returned nodes
    lb, ld, lc, ls = run(ix, q[:512], layer=True)
    lb1, ld1, lc1, ls1 = run(ix, q[3:4], layer=True)
    assert lc1[0] == lc[3] and np.array_equal(lb1[0], lb[3]) and np.array_equal(ld1[0].view(np.uint32), ld[3].view(np.uint32))
    for i in range(0, 512, 61):
        exp = np.array([L.ndbo_ivf_distance(qh[i], bh[b - 1], DIM, 1) for b in lb[i, :lc[i]]], np.float32)
        assert np.array_equal(exp.view(np.uint32), ld[i, :lc[i]].view(np.uint32)), i
    # oracle replay of a sample on the exported graph (both searches)
    vecs = np.zeros((n + 1, DIM), np.float32)
    vecs[1:] = bh
    og = ndbo.HnswGraph.from_arrays(vecs, e["levels"], e["ncount"], e["nbrs"], None, e["entry_point"],
                                    e["entry_level"], m, efc)
    for i in range(8):
        eb, ed, ns = og.search(qh[i], 2, ef, K)
        assert len(eb) == oc[i] and np.array_equal(eb, ob[i, :oc[i]]) and ns == sc[i]
        assert np.array_equal(ed.view(np.uint32), od[i, :oc[i]].view(np.uint32))
        eb, ed, ns = og.search_layer(qh[i], ef, K)
        assert np.array_equal(eb, lb[i, :lc[i]]) and np.array_equal(ed.view(np.uint32), ld[i, :lc[i]].view(np.uint32))
        assert ns == ls[i]
    ix.close()
    del og, vecs
    # the optimistic, batched build is deterministic: a second build gives the same graph
    ix2 = build()
    e2 = ix2.export()
    assert hashlib.sha256(e2["nbrs"].tobytes() + e2["ncount"].tobytes()).hexdigest() == digest
    ix2.close()


def test_c3_intended_hnsw_at_full_size():
    """the search a drop-in `ORDER BY v <=> $q` on an hnsw index runs (neurondb.ref_compat off) at C3's full size.
    1M-node `intended` graph built on the device over the bench's clustered unit rows; strategy 2 (cosine) on float4 rows
    and on fp16 walk rows: k results, ascending, every distance hnswComputeDistance(query, that node) bit for bit
    (hnsw_am.c:1321-1332), a query's answer the same alone and in a batch of 2048, recall@10 against a float64 brute
    force, and 32 queries replayed by the oracle on the exported graph for BOTH walks (blocks, float4 bits, evaluation
    counts); strategies 1 and 3: 8 queries each replayed the same way."""
    import ctypes as C
    from neurondb_amd import HnswIndex, _lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    n, m, efc, ef = N, 16, 200, 64
    base = make_data(n, DIM, "clustered", 1024, 0.1, 0x5EED0003, 0x5EEDC0DE, dev)
    base = base / base.norm(dim=1, keepdim=True)
    q = make_data(2048, DIM, "clustered", 1024, 0.1, 0x5EED0004, 0x5EEDC0DE, dev)
    q = q / q.norm(dim=1, keepdim=True)
    r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)
    ix = HnswIndex(DIM, m)
    ix.build_intended(base, torch.arange(n, device=dev, dtype=torch.int64), levels, efc)
    res = {w: ix.search_intended(q, ef, K, walk16=w, strategy=2) for w in (False, True)}
    L = ndbo.lib()
    bh, qh = base.cpu().numpy(), q.cpu().numpy()
    err = C.c_int(0)
    sims = q[:256].double() @ base.double().T
    gt = torch.topk(sims, K, dim=1).indices.cpu().numpy() + 1
    del sims
    for w, (ob, od, oc, oe) in res.items():
        assert (oc == K).all() and (np.diff(od, axis=1) >= 0).all() and (oe > ef).all()
        assert all(len(set(row.tolist())) == K for row in ob[:512])
        for i in range(0, 2048, 97):
            exp = np.array([L.ndbo_hnsw_distance(qh[i], bh[b - 1], DIM, 2, C.byref(err)) for b in ob[i]], np.float32)
            assert np.array_equal(exp.view(np.uint32), od[i].view(np.uint32)), (w, i)
        b1, d1, c1, e1 = ix.search_intended(q[9:10], ef, K, walk16=w, strategy=2)
        assert np.array_equal(b1[0], ob[9]) and np.array_equal(d1[0].view(np.uint32), od[9].view(np.uint32)) and e1[0] == oe[9]
        rec = float(np.mean([len(set(ob[i].tolist()) & set(gt[i].tolist())) / K for i in range(256)]))
        assert rec > 0.85, (w, rec)
    # the two walks agree on almost every query (they differ where halves round a row across a neighbour)
    assert (res[False][0] == res[True][0]).all(axis=1).mean() > 0.97
    e = ix.export()
    vecs = np.zeros((n + 1, DIM), np.float32)
    vecs[1:] = bh
    og = ndbo.HnswGraph.from_arrays(vecs, e["levels"], e["ncount"], e["nbrs"], None, e["entry_point"], e["entry_level"], m, efc)
    del vecs
    w16 = og.walk_rows()
    for w, (ob, od, oc, oe) in res.items():
        for i in range(32):
            eb, ed, ns = og.search_intended_s(qh[i], 2, ef, K, w16=w16 if w else None)
            assert np.array_equal(eb, ob[i, :oc[i]]) and ns == oe[i], (w, i)
            assert np.array_equal(ed.view(np.uint32), od[i, :oc[i]].view(np.uint32)), (w, i)
    for strategy in (1, 3):
        for w in (False, True):
            ob, od, oc, oe = ix.search_intended(q[:8], ef, K, walk16=w, strategy=strategy)
            for i in range(8):
                eb, ed, ns = og.search_intended_s(qh[i], strategy, ef, K, w16=w16 if w else None)
                assert np.array_equal(eb, ob[i, :oc[i]]) and ns == oe[i], (strategy, w, i)
                assert np.array_equal(ed.view(np.uint32), od[i, :oc[i]].view(np.uint32)), (strategy, w, i)
    ix.close()


def _free_gib():
    free, _ = torch.cuda.mem_get_info(0)
    return free / 2 ** 30


@pytest.mark.parametrize("cfg", ["C4", "C5"])
def test_c4_c5_single_gpu_share_properties(cfg):
    """configs[3] / configs[4] at their full row counts on ONE device (the 8-GPU runs shard these rows):
    C4 = 10M x 768 fp32, lists = 4096, probes = 32, L2;  C5 = 10M x 1536 fp16 rows, inner product, batches of 256.
    Device-only properties: sorted results, bit-identical repeats, a query's result independent of its batch,
    and every returned distance recomputed on the device by the exact-scan kernel of the same recipe
    (ndbhip_batch_distance recipe 0 on the returned rows, itself oracle-checked in test_gpu_ivf.py)."""
    from neurondb_amd import IvfIndex, _lib
    lib, check = _lib.lib(), _lib.check
    n, dim, lists, strategy, nq = (10_000_000, 768, 4096, 1, 1024) if cfg == "C4" else (10_000_000, 1536, 4096, 3, 256)
    need = n * dim * 4 / 2 ** 30 * (1.2 if cfg == "C4" else 1.8) + 8
    if _free_gib() < need:
        pytest.skip(f"needs {need:.0f} GiB of free HBM")
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    base = make_data(n, dim, "clustered", 4096, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq, dim, "clustered", 4096, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, lists)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    if cfg == "C5":
        twin = ix.to_f16(False)
        ix.close()
        ix = twin
        # what the fp16 rows decode to: fp16_to_float, whose subnormals come out 2^-10 too small (quirk Q20)
        for s0 in range(0, n, 1 << 20):
            h = base[s0:s0 + (1 << 20)].to(torch.float16)
            f = h.to(torch.float32)
            sub = (h.abs() < 2.0 ** -14) & (h != 0)
            base[s0:s0 + (1 << 20)] = torch.where(sub, f * 2.0 ** -10, f)

    def search(qs):
        m = qs.shape[0]
        ot = torch.zeros((m, K), dtype=torch.int64, device=dev)
        od = torch.zeros((m, K), dtype=torch.float32, device=dev)
        oc = torch.zeros(m, dtype=torch.int32, device=dev)
        ix.search_device(qs.contiguous(), ot, od, oc, strategy, PROBES, K, 0)
        check(lib.ndbhip_synchronize())
        return unpack_tids(ot).cpu().numpy(), od.cpu().numpy(), oc.cpu().numpy()

    rows, dist, cnt = search(q)
    assert (cnt == K).all() and (np.diff(dist, axis=1) >= 0).all()
    rows2, dist2, _ = search(q)
    assert np.array_equal(rows, rows2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32))
    for lo, m in ((0, 1), (17, 8), (100, 64)):
        r, d, _ = search(q[lo:lo + m])
        assert np.array_equal(r, rows[lo:lo + m]) and np.array_equal(d.view(np.uint32), dist[lo:lo + m].view(np.uint32))
    # distances of the returned rows, recomputed one query at a time by the exact-scan kernel
    qh = q.cpu().numpy()
    for i in range(0, nq, max(1, nq // 24)):
        v = np.ascontiguousarray(base[torch.from_numpy(rows[i]).to(dev)].cpu().numpy())
        out = np.zeros((1, K), np.float32)
        check(lib.ndbhip_batch_distance(qh[i].ctypes.data, v.ctypes.data, out.ctypes.data, 1, K, dim, strategy, 0))
        assert np.array_equal(out[0].view(np.uint32), dist[i].view(np.uint32)), i

    # Oracle replay at full size: ivfSelectClusters + ivfCollectCandidates of the CPU oracle
    # (ivf_am.c:1597-1909) for a few of the batch's queries, over an image that holds every centroid and the rows
    # of the lists those queries probe (an unprobed list is never walked, so its rows need not leave the device).
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ndbo
    cent_h, ll, _, _ = ix.export(rows=False)
    t6 = np.zeros((n, 6), np.uint8)
    check(lib.ndbhip_ivf_export(ix._h, None, None, None, t6.ctypes.data_as(C.c_void_p)))
    tid_all = t6.view(ndbo.TID_DTYPE).reshape(n)
    order = (((tid_all["bi_hi"].astype(np.int64) << 16) | tid_all["bi_lo"]) * 64 + tid_all["posid"] - 1)   # heap row of every mirror row
    off = np.zeros(len(ll) + 1, np.int64)
    off[1:] = np.cumsum(ll)
    picks, lists_u = [], set()
    for i in range(0, nq, max(1, nq // 8)):
        pr = [int(x) for x in ix.select_clusters(qh[i:i + 1], PROBES)[0] if x >= 0]
        grown = lists_u | set(pr)
        if sum(int(ll[L]) for L in grown) * dim * 4 > 16 * 2 ** 30 and picks:     # keep the host image under 16 GiB
            continue
        lists_u = grown
        picks.append(i)
    keep = np.zeros(len(ll), bool)
    keep[list(lists_u)] = True
    ll2 = np.where(keep, ll, 0)
    off2 = np.zeros(len(ll) + 1, np.int64)
    off2[1:] = np.cumsum(ll2)
    sel = np.concatenate([np.arange(off[L], off[L + 1]) for L in sorted(lists_u)]) if lists_u else np.zeros(0, np.int64)
    rows_img = np.empty((len(sel), dim), np.float32)
    for s0 in range(0, len(sel), 1 << 18):
        idx = torch.from_numpy(order[sel[s0:s0 + (1 << 18)]]).to(dev)
        rows_img[s0:s0 + (1 << 18)] = base[idx].cpu().numpy()
    img = ndbo.IvfImage(cent_h, off2, rows_img, np.ascontiguousarray(tid_all[sel]))
    with ThreadPoolExecutor(max_workers=len(picks)) as ex:
        res = list(ex.map(lambda i: img.search(qh[i], strategy, PROBES, K, 0), picks))
    for i, (et, ed, _) in zip(picks, res):
        erow = ((et["bi_hi"].astype(np.int64) << 16) | et["bi_lo"]) * 64 + et["posid"] - 1
        assert len(et) == cnt[i] and np.array_equal(erow, rows[i, :len(et)]), (cfg, i)
        assert np.array_equal(ed.view(np.uint32), dist[i, :len(et)].view(np.uint32)), (cfg, i)
    assert len(picks) >= 4

    # The 8-rank execution of this config, one rank after the other on this device: the index cut into
    # work-balanced list slices (the 900 k-row list every query probes goes to all eight), every shard's partial
    # records, the replay merge — equal to the unsharded result for the whole batch.
    from neurondb_amd.dist import ShardedSearchBuffers, partition_slices
    world = 8
    probes = torch.zeros((nq, PROBES), dtype=torch.int32, device=dev)
    ix.select_clusters_device(q, probes, PROBES)
    check(lib.ndbhip_synchronize())
    pc = probes.cpu().numpy()
    slo, sln, stl = partition_slices(ll, world, np.bincount(pc[pc >= 0].ravel(), minlength=lists)[:lists])
    assert ((sln > 0).sum(0) > 1).any()
    buf = ShardedSearchBuffers(nq, K, world, dev, nprobe=PROBES)
    for w in range(world):
        sh = ix.shard_slices(slo[w], sln[w], stl[w])
        sh.search_partial_probes_device(q, probes, buf.cand, buf.ncand, buf.total, strategy, PROBES, K, 0)
        check(lib.ndbhip_synchronize())
        buf.cand_all[w].copy_(buf.cand)
        buf.ncand_all[w].copy_(buf.ncand)
        sh.close()
    check(lib.ndbhip_merge_topk_device(buf.cand_all.data_ptr(), buf.ncand_all.data_ptr(), buf.total.data_ptr(), world, nq, K,
                                       buf.cap, buf.out_tids.data_ptr(), buf.out_dist.data_ptr(), buf.out_count.data_ptr()))
    check(lib.ndbhip_synchronize())
    assert np.array_equal(unpack_tids(buf.out_tids).cpu().numpy(), rows), cfg
    assert np.array_equal(buf.out_dist.cpu().numpy().view(np.uint32), dist.view(np.uint32)), cfg
    assert np.array_equal(buf.out_count.cpu().numpy(), cnt), cfg
    ix.close()


def test_c2_on_the_iid_table_at_full_size():
    """BASELINE.md's own C2 data — i.i.d. N(0,1) rows — at full size: the reference's build rule (k-means on the first
    10 000 rows, ivf_am.c:580) leaves most rows in a few dozen lists there, a query's 32 probes cover four fifths of the table, and
    from the second batch on the sweep runs the dense tile's kernel (csrc/ndbhip_screen16d.h; asserted from the
    statistics).  16 queries replayed in full by the CPU oracle on the exported index (ids, ranks, float4 bits), the
    batch's properties, and independence of the batch a query travels in."""
    from concurrent.futures import ThreadPoolExecutor
    from neurondb_amd import IvfIndex, _lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    base = make_data(N, DIM, "gauss", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(4096, DIM, "gauss", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(DIM, LISTS)
    ix.build_device(base, pack_tids(torch.arange(N, device=dev)), 50)
    del base
    _lib.check(_lib.lib().ndbhip_set_scan_mode(0))
    try:
        _search(ix, q)                                        # the first batch measures the pairs per bucket
        _lib.check(_lib.lib().ndbhip_stats_reset())
        rows, dist, cnt = _search(ix, q)
        st = _lib.stats()
        assert st["screen16_batches"] == 1 and st["screen16_fallbacks"] == 0, st
        assert st["dense_sweeps"] >= 1, {k: st[k] for k in ("dense_sweeps", "screen16_batches", "pairs_pruned", "rows_swept", "rows_emitted")}
        assert (cnt == K).all() and (np.diff(dist, axis=1) >= 0).all()
        assert all(len(set(r.tolist())) == K for r in rows[:512])
        # the batch a query travels in does not matter (other tiles, other thresholds, the exact path for one query)
        for lo, n in ((3, 1), (100, 40), (1000, 700)):
            r, d, c = _search(ix, q[lo:lo + n])
            assert np.array_equal(r, rows[lo:lo + n]) and np.array_equal(d.view(np.uint32), dist[lo:lo + n].view(np.uint32))
        cent, list_len, rows_h, tid_h = ix.export(rows=True)
        off = np.zeros(LISTS + 1, np.int64)
        off[1:] = np.cumsum(list_len)
        assert np.sort(list_len)[-64:].sum() > 0.5 * N        # the degenerate index the reference's rule builds here
        img = ndbo.IvfImage(cent, off, rows_h, np.ascontiguousarray(tid_h).view(ndbo.TID_DTYPE).reshape(-1))
        pick = list(range(0, 4096, 256))                      # 16 queries
        qh = q.cpu().numpy()
        ndbo.lib()
        with ThreadPoolExecutor(max_workers=8) as ex:
            res = list(ex.map(lambda i: img.search(qh[i], 1, PROBES, K, 0), pick))
        for i, (et, ed, _) in zip(pick, res):
            er = (((et["bi_hi"].astype(np.int64) << 16) | et["bi_lo"]) * 64 + et["posid"] - 1)
            assert np.array_equal(er, rows[i]) and np.array_equal(ed.view(np.uint32), dist[i].view(np.uint32)), i
    finally:
        ix.close()
