"""Host-side mirror of the reference's HNSW access-method scan interface
(NeuronDB/src/index/hnsw_am.c: hnswbeginscan/hnswrescan/hnswgettuple) over the C ABI."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from ._lib import check, ensure_init, lib

HNSW_DEFAULT_EF_SEARCH = 64   # GUC neurondb.hnsw_ef_search (src/util/neurondb_guc.c:161)
HNSW_DEFAULT_K = 10           # GUC neurondb.hnsw_k (src/util/neurondb_guc.c:174)
TID_DTYPE = np.dtype([("bi_hi", "<u2"), ("bi_lo", "<u2"), ("posid", "<u2")])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HnswIndex:
    """Device mirror of an hnsw index: node b = block number b (block 0 = meta page)."""

    def __init__(self, dim: int, m: int = 16, device: int | None = None):
        ensure_init(device)
        self.dim, self.m = int(dim), int(m)
        h = C.c_void_p()
        check(lib().ndbhip_hnsw_create(self.dim, self.m, C.byref(h)))
        self._h = h

    def close(self):
        """Destroys the handle; the shares made from it (share()) that are still open are closed first — the library refuses to
        destroy a graph with live shares, and at interpreter exit the source may well be collected before them."""
        for ref in getattr(self, "_shares", []):
            sh = ref()
            if sh is not None:
                sh.close()
        self._shares = []
        if getattr(self, "_h", None):
            check(lib().ndbhip_hnsw_destroy(self._h))
            self._h = None

    def share(self):
        """A second handle on this graph with a workspace of its own (ndbhip_hnsw_share): for a second batch of searches in
        flight on another thread and stream.  Both handles are frozen until the share is closed; close it before this one."""
        h = C.c_void_p()
        check(lib().ndbhip_hnsw_share(self._h, C.byref(h)))
        sub = HnswIndex.__new__(HnswIndex)
        sub.dim, sub.m, sub._h, sub._src = self.dim, self.m, h, self
        sub.nblocks = getattr(self, "nblocks", None)
        if not hasattr(self, "_shares"):
            self._shares = []
        self._shares.append(weakref.ref(sub))
        return sub

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, vecs, levels, ncount, nbrs_full, tids, entry_point, entry_level):
        """vecs [nb, dim]; levels [nb]; ncount [nb,16]; nbrs_full [nb,16,2m] (only levels <= level[b] are kept);
        tids [nb] structured or [nb,6] uint8."""
        nb = int(vecs.shape[0])
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        ncount = np.ascontiguousarray(ncount, dtype=np.int16)
        slots = (levels.astype(np.int64) + 1) * 2 * self.m
        slots[0] = 0
        off = np.zeros(nb + 1, dtype=np.int64)
        off[1:] = np.cumsum(slots)
        packed = np.full(int(off[-1]), 0xFFFFFFFF, dtype=np.uint32)
        nf = np.asarray(nbrs_full, dtype=np.uint32)
        for b in range(1, nb):
            packed[off[b]:off[b + 1]] = nf[b, : levels[b] + 1].reshape(-1)
        t = np.ascontiguousarray(tids)
        t6 = np.ascontiguousarray(t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6))
        check(lib().ndbhip_hnsw_load(self._h, nb, _ptr(vecs), _ptr(levels), _ptr(ncount), _ptr(off), _ptr(packed),
                                     _ptr(t6), int(entry_point) & 0xFFFFFFFF, int(entry_level)))
        self.nblocks = nb

    def build(self, rows, tids, levels, ef_construction=200):
        """hnswbuild on the device: inserts rows in order (node i+1 = row i) with the given level draws."""
        import torch
        r = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).cuda()
        t = np.ascontiguousarray(tids)
        t6 = t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6)
        t8 = np.zeros((t6.shape[0], 8), dtype=np.uint8)
        t8[:, :6] = t6
        tt = torch.from_numpy(t8.view(np.int64).reshape(-1)).cuda()
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        check(lib().ndbhip_hnsw_build_device(self._h, C.c_void_p(r.data_ptr()), C.c_void_p(tt.data_ptr()), len(lv),
                                             _ptr(lv), int(ef_construction)))
        self.nblocks = len(lv) + 1

    def build_intended(self, rows, tids, levels, ef_construction=200, batch_div=16, batch_max=8192, append=False):
        """The `intended` graph (ndbhip_hnsw_build_intended_device; oracle/ndb_oracle_hnsw2.c defines it): rows as a
        numpy array or a CUDA tensor, node i + 1 = row i.  append: the rows go on top of the graph the mirror holds
        (ndbhip_hnsw_insert_intended_device: hnswinsert under `intended`), node nblocks + i = row i."""
        import torch
        r = rows if isinstance(rows, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).cuda()
        if isinstance(tids, torch.Tensor):
            tt = tids
        else:
            t = np.ascontiguousarray(tids)
            t6 = t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6)
            t8 = np.zeros((t6.shape[0], 8), dtype=np.uint8)
            t8[:, :6] = t6
            tt = torch.from_numpy(t8.view(np.int64).reshape(-1)).cuda()
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        fn = lib().ndbhip_hnsw_insert_intended_device if append else lib().ndbhip_hnsw_build_intended_device
        check(fn(self._h, C.c_void_p(r.data_ptr()), C.c_void_p(tt.data_ptr()), len(lv),
                 _ptr(lv), int(ef_construction), int(batch_div), int(batch_max)))
        check(lib().ndbhip_synchronize())
        self.nblocks = (self.nblocks if append and getattr(self, "nblocks", 0) else 1) + len(lv)

    def search_intended(self, queries, ef=HNSW_DEFAULT_EF_SEARCH, k=HNSW_DEFAULT_K, walk16=False, strategy=1):
        """kNN of the `intended` mode: (blocks [nq, k], dist [nq, k], count [nq], evaluations [nq]).  strategy = the operator
        class's (1 L2: sqrt of the squared L2; 2 cosine, 3 negative inner product: hnswComputeDistance's float4 values).
        walk16: descent and layer search on fp16 walk rows, the result set re-scored on the float4 rows
        (ndbhip_hnsw_search_intended_w16_device)"""
        import torch
        q = queries if isinstance(queries, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)).cuda()
        nq = q.shape[0]
        ob = torch.zeros((nq, k), dtype=torch.int32, device=q.device)
        od = torch.zeros((nq, k), dtype=torch.float32, device=q.device)
        oc = torch.zeros(nq, dtype=torch.int32, device=q.device)
        oe = torch.zeros(nq, dtype=torch.int64, device=q.device)
        fn = lib().ndbhip_hnsw_search_intended_w16_device if walk16 else lib().ndbhip_hnsw_search_intended_device
        check(fn(self._h, C.c_void_p(q.data_ptr()), nq, int(strategy), int(ef), int(k), C.c_void_p(ob.data_ptr()), C.c_void_p(od.data_ptr()),
                 C.c_void_p(oc.data_ptr()), None, C.c_void_p(oe.data_ptr())))
        check(lib().ndbhip_synchronize())
        return ob.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy(), oe.cpu().numpy()

    @classmethod
    def load_pages(cls, pages):
        """Mirror of an hnsw relation image (bytes / uint8 array of 8 KB blocks): ndbhip_hnsw_load_pages."""
        ensure_init(None)
        buf = np.frombuffer(bytes(pages), np.uint8) if not isinstance(pages, np.ndarray) else pages
        nb = buf.size // 8192
        h = C.c_void_p()
        check(lib().ndbhip_hnsw_load_pages(C.byref(h), buf.ctypes.data, nb))
        d, m = C.c_int(), C.c_int()
        check(lib().ndbhip_hnsw_shape(h, C.byref(d), C.byref(m)))
        ix = cls.__new__(cls)
        ix.dim, ix.m, ix._h, ix.nblocks = d.value, m.value, h, nb
        return ix

    def write_pages(self, ef_construction=200, ef_search=64):
        """The relation image of this mirror (bytes): block 0 meta + one node page per block."""
        nb = C.c_uint32()
        check(lib().ndbhip_hnsw_write_pages(self._h, ef_construction, ef_search, None, 0, C.byref(nb)))
        pages = np.zeros(nb.value * 8192, np.uint8)
        check(lib().ndbhip_hnsw_write_pages(self._h, ef_construction, ef_search, pages.ctypes.data, nb.value,
                                            C.byref(nb)))
        return pages.tobytes()

    def insert(self, rows, tids, levels, ef_construction=200):
        """hnswinsert: more rows on top of the graph the mirror holds (node nblocks + i = row i)."""
        import torch
        r = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)).cuda()
        t = np.ascontiguousarray(tids)
        t6 = t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6)
        t8 = np.zeros((t6.shape[0], 8), dtype=np.uint8)
        t8[:, :6] = t6
        tt = torch.from_numpy(t8.view(np.int64).reshape(-1)).cuda()
        lv = np.ascontiguousarray(levels, dtype=np.int32).reshape(-1)
        check(lib().ndbhip_hnsw_insert_device(self._h, C.c_void_p(r.data_ptr()), C.c_void_p(tt.data_ptr()), len(lv),
                                              _ptr(lv), int(ef_construction)))
        self.nblocks = max(getattr(self, "nblocks", 1), 1) + len(lv)

    def delete(self, tids):
        """hnswbulkdelete: unlink and mark dead every live node whose heapPtr is in `tids`; returns the count."""
        t = np.ascontiguousarray(tids)
        t6 = np.ascontiguousarray(t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6))
        removed = C.c_int64(0)
        check(lib().ndbhip_hnsw_delete(self._h, _ptr(t6), t6.shape[0], C.byref(removed)))
        return int(removed.value)

    def build_stats(self):
        """How the last build() ran (ndbhip_hnsw_build_stats)."""
        st = np.zeros(6, np.int64)
        check(lib().ndbhip_hnsw_build_stats(self._h, _ptr(st)))
        return dict(zip(("walks", "redone", "overflowed", "rounds", "batches", "max_batch"),
                        (int(x) for x in st)))

    @staticmethod
    def set_build_mode(optimistic=True, batch_div=64, batch_max=1024):
        check(lib().ndbhip_hnsw_set_build_mode(int(optimistic), int(batch_div), int(batch_max)))

    def export(self):
        nb = C.c_uint32()
        ep = C.c_uint32()
        el = C.c_int()
        check(lib().ndbhip_hnsw_export(self._h, C.byref(nb), None, None, None, C.byref(ep), C.byref(el)))
        n = nb.value
        levels = np.zeros(n, np.int32)
        ncount = np.zeros((n, 16), np.int16)
        nbrs = np.zeros((n, 16, 2 * self.m), np.uint32)
        check(lib().ndbhip_hnsw_export(self._h, None, _ptr(levels), _ptr(ncount), _ptr(nbrs), None, None))
        return dict(levels=levels, ncount=ncount, nbrs=nbrs, entry_point=ep.value, entry_level=el.value, nblocks=n)

    def search(self, queries, strategy=1, ef=HNSW_DEFAULT_EF_SEARCH, k=HNSW_DEFAULT_K):
        """Returns (blocks [nq,k] uint32, dist [nq,k], count [nq], tids [nq,k], scored [nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ob = np.zeros((nq, k), dtype=np.uint32)
        od = np.zeros((nq, k), dtype=np.float32)
        oc = np.zeros(nq, dtype=np.int32)
        t6 = np.zeros((nq, k, 6), dtype=np.uint8)
        sc = np.zeros(nq, dtype=np.int64)
        check(lib().ndbhip_hnsw_search(self._h, _ptr(q), nq, strategy, ef, k, _ptr(ob), _ptr(od), _ptr(oc),
                                       _ptr(t6), _ptr(sc)))
        return ob, od, oc, t6.view(TID_DTYPE).reshape(nq, k), sc

    def search_layer(self, queries, ef=HNSW_DEFAULT_EF_SEARCH, k=HNSW_DEFAULT_K, strategy=1):
        """hnsw_search_layer (src/scan/hnsw_scan.c:379-477, the reference's unused best-first search): same
        outputs as search(), results in slot order; `strategy` is accepted and ignored like the reference's."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        ob = np.zeros((nq, k), dtype=np.uint32)
        od = np.zeros((nq, k), dtype=np.float32)
        oc = np.zeros(nq, dtype=np.int32)
        t6 = np.zeros((nq, k, 6), dtype=np.uint8)
        sc = np.zeros(nq, dtype=np.int64)
        check(lib().ndbhip_hnsw_search_layer(self._h, _ptr(q), nq, strategy, ef, k, _ptr(ob), _ptr(od), _ptr(oc),
                                             _ptr(t6), _ptr(sc)))
        return ob, od, oc, t6.view(TID_DTYPE).reshape(nq, k), sc


class HnswScan:
    """hnswbeginscan / hnswrescan / hnswgettuple / hnswendscan (hnsw_am.c:880-1084)."""

    def __init__(self, index: HnswIndex, ef_search=HNSW_DEFAULT_EF_SEARCH, k=HNSW_DEFAULT_K):
        self.index = index
        self.ef_search, self.k = ef_search, k     # the two GUCs read in hnswrescan (:923-936, 974)
        self.query = None
        self.strategy = 1
        self.first_call = True
        self.results = self.distances = self.tids = None
        self.result_count = 0
        self.current = 0
        self.xs_heaptid = None

    def rescan(self, query, strategy=1):
        self.query = None if query is None else np.array(query, dtype=np.float32, copy=True)
        self.strategy = strategy                    # orderbys[0].sk_strategy (:918-921)
        self.first_call = True
        self.result_count = self.current = 0

    def gettuple(self) -> bool:
        if self.query is None:
            return False
        if self.first_call:                          # :978-1007
            b, d, c, t, _ = self.index.search(self.query[None, :], self.strategy, self.ef_search, self.k)
            self.results, self.distances, self.tids = b[0], d[0], t[0]
            self.result_count = int(c[0])
            self.first_call = False
            self.current = 0
        if self.current < self.result_count:         # :1009-1053: node->heapPtr -> xs_heaptid (no orderbyvals: Q13)
            self.xs_heaptid = self.tids[self.current]
            self.current += 1
            return True
        return False

    def endscan(self):
        self.query = self.results = self.distances = self.tids = None
