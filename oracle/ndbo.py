"""ctypes binding of the CPU oracle (oracle/ndb_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package (neurondb_amd/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

TID_DTYPE = np.dtype([("bi_hi", "<u2"), ("bi_lo", "<u2"), ("posid", "<u2")])


class NdboTid(C.Structure):
    _fields_ = [("bi_hi", C.c_uint16), ("bi_lo", C.c_uint16), ("posid", C.c_uint16)]


class NdboIvf(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("nlists", C.c_int), ("maxoff", C.c_int),
        ("centroids", C.c_void_p), ("centroid_dim", C.c_void_p),
        ("list_off", C.c_void_p), ("vecs", C.c_void_p), ("tids", C.c_void_p),
        ("live", C.c_void_p),
    ]


class NdboHnsw(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("m", C.c_int), ("ef_construction", C.c_int),
        ("entry_point", C.c_uint32), ("entry_level", C.c_int), ("max_level", C.c_int),
        ("inserted", C.c_int64), ("nblocks", C.c_uint32), ("cap_blocks", C.c_uint32),
        ("vecs", C.POINTER(C.c_float)), ("heap_tids", C.POINTER(NdboTid)),
        ("levels", C.POINTER(C.c_int)), ("ncount", C.POINTER(C.c_int16)),
        ("nbrs", C.POINTER(C.c_uint32)), ("dead", C.POINTER(C.c_uint8)),
    ]


def build(native: bool = False) -> str:
    """Compile the oracle with oracle/Makefile; returns the .so path."""
    target = "libndboracle_native.so" if native else "libndboracle.so"
    path = os.path.join(_BUILD, target)
    src = os.path.join(_HERE, "ndb_oracle.c")
    hdr = os.path.join(_HERE, "ndb_oracle.h")
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in (src, hdr))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, os.path.join("_build", target)],
                              stdout=subprocess.DEVNULL)
    return path


_LIBS: dict = {}


def lib(native: bool = False):
    key = bool(native)
    if key in _LIBS:
        return _LIBS[key]
    L = C.CDLL(build(native))
    f, i, p = C.c_float, C.c_int, C.c_void_p
    sig = {
        "ndbo_ivf_distance": (f, [f32p, f32p, i, i]),
        "ndbo_ivf_l2sq": (f, [f32p, f32p, i]),
        "ndbo_hnsw_distance": (f, [f32p, f32p, i, i, C.POINTER(C.c_int)]),
        "ndbo_op_l2_scalar": (f, [f32p, f32p, i]),
        "ndbo_op_ip_scalar": (f, [f32p, f32p, i]),
        "ndbo_op_cosine_scalar": (f, [f32p, f32p, i]),
        "ndbo_op_l2_simd": (f, [f32p, f32p, i, i]),
        "ndbo_op_ip_simd": (f, [f32p, f32p, i, i]),
        "ndbo_op_cosine_simd": (f, [f32p, f32p, i, i]),
        "ndbo_op_l2": (f, [f32p, f32p, i, i]),
        "ndbo_op_ip": (f, [f32p, f32p, i, i]),
        "ndbo_op_cosine": (f, [f32p, f32p, i, i]),
        "ndbo_hnsw_bulkdelete": (C.c_int64, [C.POINTER(NdboHnsw), C.c_void_p, C.c_int64]),
        "ndbo_float4_to_fp16": (C.c_uint16, [f]),
        "ndbo_fp16_to_float": (f, [C.c_uint16]),
        "ndbo_halfvec_l2": (f, [u16p, u16p, i]),
        "ndbo_halfvec_cosine": (f, [u16p, u16p, i]),
        "ndbo_halfvec_ip": (f, [u16p, u16p, i]),
        "ndbo_ivf_select_clusters": (i, [C.POINTER(NdboIvf), f32p, i, i32p]),
        "ndbo_ivf_collect_candidates": (i, [C.POINTER(NdboIvf), f32p, i, i32p, i, i, C.c_int64,
                                             p, f32p, C.POINTER(C.c_int64)]),
        "ndbo_ivf_search": (i, [C.POINTER(NdboIvf), f32p, i, i, i, C.c_int64, p, f32p,
                                 C.POINTER(C.c_int64)]),
        "ndbo_kmeans": (i, [f32p, i, i, i, i, f, f32p, i32p, i32p, C.POINTER(f)]),
        "ndbo_kmeans_assign": (None, [f32p, i, i, f32p, i, i32p, i32p]),
        "ndbo_kmeans_update": (None, [f32p, i, i, i32p, i32p, i, f32p]),
        "ndbo_kmeans_cost": (f, [f32p, i, i, i32p, f32p]),
        "ndbo_ivf_assign": (i, [f32p, p, i, i, i, f32p, C.POINTER(f)]),
        "ndbo_hnsw_create": (C.POINTER(NdboHnsw), [i, i, i, C.c_uint32]),
        "ndbo_hnsw_free": (None, [C.POINTER(NdboHnsw)]),
        "ndbo_hnsw_search_layer": (i, [C.POINTER(NdboHnsw), f32p, i, i, u32p, f32p, C.POINTER(C.c_int64)]),
        "ndbo_hnsw_search": (i, [C.POINTER(NdboHnsw), f32p, i, i, i, u32p, f32p,
                                  C.POINTER(C.c_int64)]),
        "ndbo_hnsw_insert": (C.c_uint32, [C.POINTER(NdboHnsw), f32p, NdboTid, i]),
        "ndbo_hnsw_level_from_uniform": (i, [C.c_double, f]),
        "ndbo_selection_topk": (i, [f32p, C.c_int64, i, i64p]),
        "ndbo_h2_dist2": (C.c_double, [f32p, f32p, i]),
        "ndbo_h2_build": (i, [C.POINTER(NdboHnsw), f32p, C.c_void_p, C.c_int64, i32p, i, i, i]),
        "ndbo_h2_search": (i, [C.POINTER(NdboHnsw), f32p, i, i, u32p, f32p, C.POINTER(C.c_int64)]),
        "ndbo_h2_walk_rows": (None, [f32p, C.c_int64, u16p]),
        "ndbo_h2_dist2_w16": (C.c_double, [f32p, u16p, i]),
        "ndbo_h2_search_w16": (i, [C.POINTER(NdboHnsw), u16p, f32p, i, i, u32p, f32p, C.POINTER(C.c_int64)]),
        "ndbo_h2_search_s": (i, [C.POINTER(NdboHnsw), C.c_void_p, i, f32p, i, i, u32p, f32p, C.POINTER(C.c_int64)]),
        "ndbo_h2_rinv": (C.c_double, [C.POINTER(NdboHnsw), C.c_void_p, C.c_uint32]),
        "ndbo_h2_walk_key": (C.c_double, [C.POINTER(NdboHnsw), C.c_void_p, f32p, C.c_uint32, i]),
        "ndbo_mt_spread_copy": (C.c_void_p, [C.c_void_p, C.c_size_t, i]),
        "ndbo_mt_ivf_search_batch": (C.c_double, [C.POINTER(NdboIvf), f32p, i, i, i, i, C.c_int64, i, C.c_void_p, f32p,
                                                  i32p, C.POINTER(C.c_int64)]),
        "ndbo_mt_ivf_assign_batch": (C.c_double, [C.c_void_p, C.c_int64, i, f32p, i, i, i32p]),
        "ndbo_extract_vector": (i, [i, u8p, f32p, C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIBS[key] = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class IvfImage:
    """Flat IVF image handed to the oracle (see ndbo_ivf in ndb_oracle.h)."""

    def __init__(self, centroids, list_off, vecs, tids, nlists=None, live=None, centroid_dim=None):
        self.centroids = _f32(centroids)
        self.vecs = _f32(vecs)
        self.dim = int(self.centroids.shape[1])
        self.maxoff = int(self.centroids.shape[0])
        self.nlists = int(nlists if nlists is not None else self.maxoff)
        self.list_off = np.ascontiguousarray(list_off, dtype=np.int64)
        self.tids = np.ascontiguousarray(tids, dtype=TID_DTYPE)
        self.live = None if live is None else np.ascontiguousarray(live, dtype=np.uint8)
        self.centroid_dim = None if centroid_dim is None else np.ascontiguousarray(centroid_dim, np.int32)
        s = NdboIvf()
        s.dim, s.nlists, s.maxoff = self.dim, self.nlists, self.maxoff
        s.centroids = self.centroids.ctypes.data
        s.centroid_dim = None if self.centroid_dim is None else self.centroid_dim.ctypes.data
        s.list_off = self.list_off.ctypes.data
        s.vecs = self.vecs.ctypes.data if self.vecs.size else None
        s.tids = self.tids.ctypes.data if self.tids.size else None
        s.live = None if self.live is None else self.live.ctypes.data
        self.c = s

    def search_batch_mt(self, queries, strategy=1, nprobe=10, k=10, max_candidates=0, nthreads=1, native=False,
                        spread=True):
        """ndbo_mt_ivf_search_batch: the queries through ndbo_ivf_search on `nthreads` pthreads (one backend per core).
        spread: the rows are first copied into memory whose pages the worker threads touch first (2 MiB stripes
        round-robin), so a many-socket host reads them from all its memory controllers.
        Returns (tids [nq, k] structured, dist [nq, k], count [nq], wall seconds of the parallel section)."""
        import ctypes.util
        L = lib(native)
        q = _f32(queries)
        nq = len(q)
        c = self.c
        if spread and self.vecs.size and getattr(self, "_spread_threads", 0) != nthreads:
            self.free_spread()
            self._spread_ptr = L.ndbo_mt_spread_copy(self.vecs.ctypes.data, self.vecs.nbytes, nthreads)
            self._spread_threads = nthreads if self._spread_ptr else 0
        if spread and getattr(self, "_spread_ptr", None):
            c = NdboIvf()
            for f, _ in NdboIvf._fields_:
                setattr(c, f, getattr(self.c, f))
            c.vecs = self._spread_ptr
        out_t = np.zeros((nq, max(k, 1)), dtype=TID_DTYPE)
        out_d = np.zeros((nq, max(k, 1)), dtype=np.float32)
        out_c = np.zeros(nq, dtype=np.int32)
        ns = C.c_int64(0)
        wall = L.ndbo_mt_ivf_search_batch(C.byref(c), q, nq, strategy, nprobe, k, int(max_candidates), int(nthreads),
                                          out_t.ctypes.data, out_d, out_c, C.byref(ns))
        return out_t, out_d, out_c, float(wall)

    def free_spread(self):
        if getattr(self, "_spread_ptr", None):
            C.CDLL(None).free(C.c_void_p(self._spread_ptr))
            self._spread_ptr = None
            self._spread_threads = 0

    def select_clusters(self, query, nprobe, native=False):
        sel = np.zeros(max(nprobe, 1), dtype=np.int32)
        lib(native).ndbo_ivf_select_clusters(C.byref(self.c), _f32(query), nprobe, sel)
        return sel[:nprobe]

    def search(self, query, strategy=1, nprobe=10, k=10, max_candidates=0, native=False):
        """Returns (tids[k] structured, dist[k], n_scored)."""
        out_t = np.zeros(max(k, 1), dtype=TID_DTYPE)
        out_d = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        n = lib(native).ndbo_ivf_search(C.byref(self.c), _f32(query), strategy, nprobe, k,
                                        int(max_candidates), out_t.ctypes.data, out_d, C.byref(ns))
        return out_t[:n], out_d[:n], ns.value


def hnsw_distance(a, b, strategy):
    """ndbo_hnsw_distance = hnswComputeDistance (hnsw_am.c:1301-1345) of two float4 vectors"""
    a, b = _f32(a), _f32(b)
    err = C.c_int(0)
    d = lib().ndbo_hnsw_distance(a, b, len(a), int(strategy), C.byref(err))
    if err.value:
        raise ValueError(f"hnsw: unsupported distance strategy {strategy}")
    return np.float32(d)


def tids_from_rows(rows):
    """Synthetic heap TIDs for row numbers: block = row // 64, offset = row % 64 + 1."""
    rows = np.asarray(rows, dtype=np.int64)
    t = np.zeros(rows.shape, dtype=TID_DTYPE)
    blk = rows // 64
    t["bi_hi"] = (blk >> 16) & 0xFFFF
    t["bi_lo"] = blk & 0xFFFF
    t["posid"] = rows % 64 + 1
    return t


def tids_to_u64(t):
    """Pack structured TIDs into uint64 (bi_hi<<32 | bi_lo<<16 | posid) for comparisons."""
    t = np.asarray(t)
    return (t["bi_hi"].astype(np.uint64) << np.uint64(32)) | \
           (t["bi_lo"].astype(np.uint64) << np.uint64(16)) | t["posid"].astype(np.uint64)


def tids_from_device_u64(x):
    """Device TID format (uint64 = little-endian ItemPointerData image) -> comparable uint64."""
    x = np.ascontiguousarray(x).astype(np.uint64)
    b = x.view(np.uint8).reshape(x.shape + (8,))[..., :6]
    t = np.ascontiguousarray(b).view(TID_DTYPE).reshape(x.shape)
    return tids_to_u64(t)


def kmeans(data, k, max_iter=50, threshold=0.001, native=False):
    data = _f32(data)
    n, dim = data.shape
    cent = np.zeros((k, dim), dtype=np.float32)
    asg = np.zeros(n, dtype=np.int32)
    cnt = np.zeros(k, dtype=np.int32)
    cost = C.c_float(0)
    iters = lib(native).ndbo_kmeans(data, n, dim, k, max_iter, np.float32(threshold), cent, asg, cnt,
                                    C.byref(cost))
    return cent, asg, cnt, iters, cost.value


def ivf_assign_all(centroids, vecs, native=False):
    """Insert-time assignment (ivf_am.c:905-935) of every row; returns list ids."""
    centroids = _f32(centroids)
    vecs = _f32(vecs)
    L = lib(native)
    nl, dim = centroids.shape
    out = np.empty(len(vecs), dtype=np.int32)
    for r in range(len(vecs)):
        out[r] = L.ndbo_ivf_assign(centroids, None, nl, nl, dim, vecs[r], None)
    return out


def build_ivf_image(base, nlists, max_iter=50, native=False):
    """Intended-mode IVF build: sample first min(10000, 100*nlists) rows,
    k-means, assign every row with the insert-time rule, lists in insertion order."""
    base = _f32(base)
    n = len(base)
    ns = min(10000, nlists * 100, n)
    cent, _, _, iters, cost = kmeans(base[:ns], nlists, max_iter=max_iter, native=native)
    asg = ivf_assign_all(cent, base, native=native)
    order = np.argsort(asg, kind="stable")
    counts = np.bincount(asg, minlength=nlists)
    off = np.zeros(nlists + 1, dtype=np.int64)
    off[1:] = np.cumsum(counts)
    return IvfImage(cent, off, base[order], tids_from_rows(order)), asg, iters


class HnswGraph:
    def __init__(self, dim, m=16, ef_construction=200, cap_nodes=1024, native=False):
        self.L = lib(native)
        self.g = self.L.ndbo_hnsw_create(dim, m, ef_construction, cap_nodes)
        self.dim, self.m, self.cap = dim, m, cap_nodes

    def __del__(self):
        try:
            self.L.ndbo_hnsw_free(self.g)
        except Exception:
            pass

    @classmethod
    def from_arrays(cls, vecs, levels, ncount, nbrs, tids, entry_point, entry_level, m, ef_construction=200,
                    native=False, cap_nodes=None):
        """Oracle graph image filled from dense arrays (node b = row b of vecs, row 0 unused);
        cap_nodes = room for later inserts."""
        nb, dim = vecs.shape
        g = cls(dim, m=m, ef_construction=ef_construction, cap_nodes=max(nb, cap_nodes or 0), native=native)
        s = g.g.contents
        C.memmove(s.vecs, np.ascontiguousarray(vecs, np.float32).ctypes.data, nb * dim * 4)
        C.memmove(s.levels, np.ascontiguousarray(levels, np.int32).ctypes.data, nb * 4)
        C.memmove(s.ncount, np.ascontiguousarray(ncount, np.int16).ctypes.data, nb * 16 * 2)
        C.memmove(s.nbrs, np.ascontiguousarray(nbrs, np.uint32).ctypes.data, nb * 16 * 2 * m * 4)
        if tids is not None:
            C.memmove(s.heap_tids, np.ascontiguousarray(tids).ctypes.data, nb * 6)
        s.nblocks = nb
        s.entry_point = int(entry_point) & 0xFFFFFFFF
        s.entry_level = int(entry_level)
        s.inserted = nb - 1
        return g

    def insert(self, vec, row, level):
        t = tids_from_rows(np.array([row]))[0]
        return self.L.ndbo_hnsw_insert(self.g, _f32(vec), NdboTid(int(t["bi_hi"]), int(t["bi_lo"]),
                                                                   int(t["posid"])), int(level))

    def bulkdelete(self, tids):
        """hnswbulkdelete with callback = membership in `tids` (structured TID array); returns tuples_removed."""
        t = np.ascontiguousarray(tids)
        return int(self.L.ndbo_hnsw_bulkdelete(self.g, t.ctypes.data, len(t)))

    def build_intended(self, vecs, levels, tids=None, batch_div=64, batch_max=1024, select=0):
        """ndbo_h2_build (oracle/ndb_oracle_hnsw2.c): the `intended` graph over all rows at once, batch schedule
        clamp(nodes so far / batch_div, 1, batch_max); returns the number of batches"""
        v = _f32(vecs)
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        t = None if tids is None else np.ascontiguousarray(tids).ctypes.data
        return int(self.L.ndbo_h2_build(self.g, v, t, len(v), lv, int(batch_div), int(batch_max), int(select)))

    def search_intended(self, query, ef=64, k=10):
        ob = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        n = self.L.ndbo_h2_search(self.g, _f32(query), ef, k, ob, od, C.byref(ns))
        return ob[:n], od[:n], ns.value

    def walk_rows(self):
        """ndbo_h2_walk_rows over the graph's vectors: [nblocks, dim] uint16 (the reference's float4_to_fp16 of every element)"""
        nb, dim = int(self.g.contents.nblocks), int(self.g.contents.dim)
        v = np.ctypeslib.as_array(self.g.contents.vecs, shape=(nb * dim,))
        out = np.zeros(nb * dim, dtype=np.uint16)
        self.L.ndbo_h2_walk_rows(np.ascontiguousarray(v, dtype=np.float32), nb * dim, out)
        return out.reshape(nb, dim)

    def search_intended_w16(self, w16, query, ef=64, k=10):
        """ndbo_h2_search_w16: walk on the fp16 walk rows, the final result set re-scored on the float4 rows"""
        ob = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        w = np.ascontiguousarray(w16, dtype=np.uint16)
        n = self.L.ndbo_h2_search_w16(self.g, w.reshape(-1), _f32(query), ef, k, ob, od, C.byref(ns))
        return ob[:n], od[:n], ns.value

    def search_intended_s(self, query, strategy, ef=64, k=10, w16=None):
        """ndbo_h2_search_s: the intended search under the operator class's strategy (1 L2, 2 cosine, 3 negative inner
        product); w16 (walk_rows()) = walk on the fp16 walk rows"""
        ob = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        w = None if w16 is None else np.ascontiguousarray(w16, dtype=np.uint16)
        n = self.L.ndbo_h2_search_s(self.g, None if w is None else w.ctypes.data, int(strategy), _f32(query), ef, k, ob, od, C.byref(ns))
        if n < 0:
            raise ValueError(f"hnsw: unsupported distance strategy {strategy}")
        return ob[:n], od[:n], ns.value

    def search(self, query, strategy=1, ef=64, k=10):
        ob = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        n = self.L.ndbo_hnsw_search(self.g, _f32(query), strategy, ef, k, ob, od, C.byref(ns))
        return ob[:n], od[:n], ns.value

    def search_layer(self, query, ef=64, k=10):
        """src/scan/hnsw_scan.c: hnsw_search_layer (best-first, compute_l2_distance); results in slot order"""
        ob = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        ns = C.c_int64(0)
        n = self.L.ndbo_hnsw_search_layer(self.g, _f32(query), ef, k, ob, od, C.byref(ns))
        return ob[:n], od[:n], ns.value

    # views of the graph arrays (for loading the device mirror)
    def arrays(self):
        g = self.g.contents
        nb = g.nblocks
        dim, m = g.dim, g.m
        vecs = np.ctypeslib.as_array(g.vecs, shape=(g.cap_blocks * dim,))[: nb * dim].reshape(nb, dim).copy()
        levels = np.ctypeslib.as_array(g.levels, shape=(g.cap_blocks,))[:nb].copy()
        ncount = np.ctypeslib.as_array(g.ncount, shape=(g.cap_blocks * 16,))[: nb * 16].reshape(nb, 16).copy()
        nbrs = np.ctypeslib.as_array(g.nbrs, shape=(g.cap_blocks * 16 * 2 * m,))[: nb * 16 * 2 * m] \
            .reshape(nb, 16, 2 * m).copy()
        tids = np.ctypeslib.as_array(C.cast(g.heap_tids, C.POINTER(C.c_uint16)),
                                     shape=(g.cap_blocks * 3,))[: nb * 3].reshape(nb, 3).copy()
        return dict(vecs=vecs, levels=levels, ncount=ncount, nbrs=nbrs, tids=tids,
                    entry_point=int(g.entry_point), entry_level=int(g.entry_level), nblocks=int(nb),
                    m=int(m), dim=int(dim))
