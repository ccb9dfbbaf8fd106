"""Host-side mirror of the reference's IVF access-method interface
(NeuronDB/src/index/ivf_am.c) over the C ABI.

IvfIndex  ~ the index relation: centroids + inverted lists, resident in HBM.
IvfScan   ~ IndexScanDesc + IvfScanOpaqueData: rescan(query) / gettuple(),
            same state machine as ivfrescan / ivfgettuple (ivf_am.c:1439-1545,
            1911-2027): the first gettuple() call does all the work, later
            calls stream the buffered (tid, distance) pairs.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import check, ensure_init, lib

IVF_DEFAULT_NPROBE = 10   # ivf_am.c:55
IVF_DEFAULT_K = 10        # ivf_am.c:1543

TID_DTYPE = np.dtype([("bi_hi", "<u2"), ("bi_lo", "<u2"), ("posid", "<u2")])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class IvfIndex:
    def __init__(self, dim: int, nlists: int, device: int | None = None):
        ensure_init(device)
        self.dim, self.nlists = int(dim), int(nlists)
        h = C.c_void_p()
        check(lib().ndbhip_ivf_create(self.dim, self.nlists, C.byref(h)))
        self._h = h
        self._keep = []          # device tensors adopted by load_device

    def close(self):
        """Destroys the handle; the shares made from it (share()) that are still open are closed first — the library refuses to
        destroy a mirror with live shares, and at interpreter exit the source may well be collected before them."""
        for ref in getattr(self, "_shares", []):
            sh = ref()
            if sh is not None:
                sh.close()
        self._shares = []
        if getattr(self, "_h", None):
            check(lib().ndbhip_ivf_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- loading ---------------------------------------------------------
    def set_centroids(self, centroids):
        c = np.ascontiguousarray(centroids, dtype=np.float32)
        assert c.ndim == 2 and c.shape[1] == self.dim
        check(lib().ndbhip_ivf_set_centroids(self._h, _ptr(c), c.shape[0]))
        self.ncent = c.shape[0]

    def load(self, list_len, rows, tids, owned=None):
        """rows: [n, dim] float32 of the OWNED lists, list-major; tids: structured TID array or [n,6] uint8."""
        ll = np.ascontiguousarray(list_len, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)
        t = np.ascontiguousarray(tids)
        t6 = t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6)
        t6 = np.ascontiguousarray(t6)
        ow = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        check(lib().ndbhip_ivf_load(self._h, _ptr(ll), _ptr(ow), _ptr(rows), _ptr(t6), rows.shape[0]))

    def load_f16(self, list_len, rows_f16, tids, owned=None):
        """halfvec column: rows as uint16 fp16 images [n, dim] of the OWNED lists, list-major."""
        ll = np.ascontiguousarray(list_len, dtype=np.int64)
        rows = np.ascontiguousarray(rows_f16, dtype=np.uint16).reshape(-1, self.dim)
        t = np.ascontiguousarray(tids)
        t6 = np.ascontiguousarray(t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6))
        ow = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        check(lib().ndbhip_ivf_load_f16(self._h, _ptr(ll), _ptr(ow), _ptr(rows), _ptr(t6), rows.shape[0]))

    def load_device(self, list_len, d_rows, d_tids, owned=None):
        """d_rows: torch float32 [n, dim] on the device, d_tids: torch int64 [n] (device TID format)."""
        ll = np.ascontiguousarray(list_len, dtype=np.int64)
        ow = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        assert d_rows.is_contiguous() and d_tids.is_contiguous()
        check(lib().ndbhip_ivf_load_device(self._h, _ptr(ll), _ptr(ow), C.c_void_p(d_rows.data_ptr()),
                                           C.c_void_p(d_tids.data_ptr()), d_rows.shape[0]))
        self._keep = [d_rows, d_tids]

    def append(self, list_id, vec, tid):
        """aminsert: add one entry at the tail of list `list_id` (ivf_am.c:954-1157)."""
        v = np.ascontiguousarray(vec, dtype=np.float32)
        t = np.ascontiguousarray(np.asarray(tid).reshape(1)).view(np.uint8).reshape(-1)[:6].copy()
        check(lib().ndbhip_ivf_append(self._h, int(list_id), _ptr(v), _ptr(t)))

    def build_device(self, d_rows, d_tids, max_iter=50):
        """ambuild on data already in HBM: sample the first min(10000, 100*nlists) rows, k-means
        (ivf_am.c:2070-2294), assign every row (ivf_am.c:905-935), pack lists in heap order.
        d_rows torch float32 [n, dim], d_tids torch int64 [n]. Returns Lloyd iterations run."""
        iters = C.c_int(0)
        assert d_rows.is_contiguous() and d_tids.is_contiguous()
        check(lib().ndbhip_ivf_build_device(self._h, C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_tids.data_ptr()),
                                            d_rows.shape[0], max_iter, C.byref(iters)))
        self.ncent = self.nlists
        return iters.value

    def prepare(self, strategy=1):
        """ndbhip_ivf_prepare: lay out now what the first batched scan would prepare lazily (sublists, fp16 planes,
        norms, radii), so the index is searchable at full speed when the build returns."""
        check(lib().ndbhip_ivf_prepare(self._h, int(strategy)))

    def build_sharded_device(self, d_rows, d_tids, max_iter=50):
        """ambuild over the ranks of the library's communicator (ndbhip_ivf_build_sharded): this rank passes its
        contiguous slice of the table in heap order and ends up holding its own lists' rows.  Returns (Lloyd
        iterations, owned flags [nlists] uint8)."""
        iters = C.c_int(0)
        owned = np.zeros(self.nlists, dtype=np.uint8)
        assert d_rows.is_contiguous() and d_tids.is_contiguous()
        check(lib().ndbhip_ivf_build_sharded(self._h, C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_tids.data_ptr()),
                                             d_rows.shape[0], max_iter, C.byref(iters), _ptr(owned)))
        self.ncent = self.nlists
        return iters.value, owned

    def build(self, rows, tids, max_iter=50):
        """ndbhip_ivf_build: ivfbuild for a table in host memory (heap order); the library uploads it through its
        pinned lanes while the k-means on the sample runs.  tids: ndbo.TID_DTYPE / (n, 6) uint8 / uint64 images."""
        r = np.ascontiguousarray(rows, dtype=np.float32)
        t = np.ascontiguousarray(tids)
        if t.dtype in (np.uint64, np.int64):
            t6 = np.ascontiguousarray(t.view(np.uint8).reshape(-1, 8)[:, :6])
        else:
            t6 = t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6)
        it = C.c_int(0)
        check(lib().ndbhip_ivf_build(self._h, _ptr(r), _ptr(t6), r.shape[0], max_iter, C.byref(it)))
        check(lib().ndbhip_synchronize())
        return it.value

    def shard(self, owned):
        """New IvfIndex holding only the lists with owned[L] != 0 (multi-GPU: one process per GPU)."""
        ow = np.ascontiguousarray(owned, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().ndbhip_ivf_shard(self._h, _ptr(ow), C.byref(h)))
        sub = IvfIndex.__new__(IvfIndex)
        sub.dim, sub.nlists, sub._h, sub._keep = self.dim, self.nlists, h, []
        sub.ncent = getattr(self, "ncent", self.nlists)
        return sub

    def delete(self, tids):
        """ambulkdelete: drop every entry whose heapPtr is in `tids` (structured TID array or [n, 6] bytes);
        returns the number of entries removed."""
        t = np.ascontiguousarray(tids)
        t6 = np.ascontiguousarray(t.view(np.uint8).reshape(-1, 6) if t.dtype != np.uint8 else t.reshape(-1, 6))
        removed = C.c_int64(0)
        check(lib().ndbhip_ivf_delete(self._h, _ptr(t6), t6.shape[0], C.byref(removed)))
        return int(removed.value)

    def share(self):
        """A second handle on this mirror with scratch of its own (ndbhip_ivf_share): for a second batch in flight on another
        thread and stream.  Both handles are frozen (no loads, appends, deletes, builds) until the share is closed; close
        the shares before this index."""
        h = C.c_void_p()
        check(lib().ndbhip_ivf_share(self._h, C.byref(h)))
        sub = IvfIndex.__new__(IvfIndex)
        sub.dim, sub.nlists, sub._h, sub._keep = self.dim, self.nlists, h, [self]
        sub.ncent = getattr(self, "ncent", self.nlists)
        if not hasattr(self, "_shares"):
            self._shares = []
        self._shares.append(weakref.ref(sub))
        return sub

    def to_f16(self, reference_encoder=True):
        """Halfvec twin of this float4 mirror (rows narrowed on the device; ndbhip_ivf_to_f16)."""
        h = C.c_void_p()
        check(lib().ndbhip_ivf_to_f16(self._h, int(bool(reference_encoder)), C.byref(h)))
        sub = IvfIndex.__new__(IvfIndex)
        sub.dim, sub.nlists, sub._h, sub._keep = self.dim, self.nlists, h, []
        sub.ncent = getattr(self, "ncent", self.nlists)
        return sub

    def shard_slices(self, lo, length, tail=None):
        """New IvfIndex holding positions [lo[L], lo[L] + length[L]) of every list L (ndbhip_ivf_shard_slices);
        tail[L] != 0 = this shard takes later appends to list L."""
        a = np.ascontiguousarray(lo, dtype=np.int64)
        b = np.ascontiguousarray(length, dtype=np.int64)
        t = None if tail is None else np.ascontiguousarray(tail, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().ndbhip_ivf_shard_slices(self._h, _ptr(a), _ptr(b), None if t is None else _ptr(t), C.byref(h)))
        sub = IvfIndex.__new__(IvfIndex)
        sub.dim, sub.nlists, sub._h, sub._keep = self.dim, self.nlists, h, []
        sub.ncent = getattr(self, "ncent", self.nlists)
        return sub

    def export(self, rows=True):
        """Read the mirror back: (centroids, list_len, rows, tids structured)."""
        nc = lib().ndbhip_ivf_ncentroids(self._h)
        n = self.nrows
        cent = np.zeros((nc, self.dim), dtype=np.float32)
        ll = np.zeros(nc, dtype=np.int64)
        r = np.zeros((n, self.dim), dtype=np.float32) if rows else None
        t6 = np.zeros((n, 6), dtype=np.uint8) if rows else None
        check(lib().ndbhip_ivf_export(self._h, _ptr(cent), _ptr(ll), _ptr(r), _ptr(t6)))
        return cent, ll, r, (None if t6 is None else t6.view(TID_DTYPE).reshape(n))

    @property
    def nrows(self):
        return lib().ndbhip_ivf_nrows(self._h)

    def max_candidates(self, nprobe):
        return lib().ndbhip_ivf_max_candidates(self._h, nprobe)

    # -- search ----------------------------------------------------------
    def search(self, queries, strategy=1, nprobe=IVF_DEFAULT_NPROBE, k=IVF_DEFAULT_K, max_candidates=0):
        """Host arrays in/out. Returns (tids [nq,k] structured, dist [nq,k], count [nq])."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        t6 = np.zeros((nq, k, 6), dtype=np.uint8)
        d = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        check(lib().ndbhip_ivf_search(self._h, _ptr(q), nq, strategy, nprobe, k, int(max_candidates),
                                      _ptr(t6), _ptr(d), _ptr(cnt)))
        return t6.view(TID_DTYPE).reshape(nq, k), d, cnt

    def search_mapped(self, base_ptr, offsets, strategy=1, nprobe=IVF_DEFAULT_NPROBE, k=IVF_DEFAULT_K, max_candidates=0):
        """Queries scattered in device-visible host memory: query i = the dim floats at base_ptr + offsets[i] (bytes);
        base_ptr = the DEVICE pointer of that memory (ndbhip_ivf_search_mapped: what the device-owner service does with
        its request ring).  Returns like search()."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        nq = len(off)
        t6 = np.zeros((nq, k, 6), dtype=np.uint8)
        d = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.int32)
        check(lib().ndbhip_ivf_search_mapped(self._h, C.c_void_p(int(base_ptr)), _ptr(off), nq, strategy, nprobe, k,
                                             int(max_candidates), _ptr(t6), _ptr(d), _ptr(cnt)))
        return t6.view(TID_DTYPE).reshape(nq, k), d, cnt

    def search_device(self, d_queries, out_tids, out_dist, out_count, strategy=1, nprobe=IVF_DEFAULT_NPROBE,
                      k=IVF_DEFAULT_K, max_candidates=0):
        """torch device tensors in/out; asynchronous on the current ndbhip stream."""
        nq = d_queries.shape[0]
        check(lib().ndbhip_ivf_search_device(self._h, C.c_void_p(d_queries.data_ptr()), nq, strategy, nprobe, k,
                                             int(max_candidates), C.c_void_p(out_tids.data_ptr()),
                                             C.c_void_p(out_dist.data_ptr()), C.c_void_p(out_count.data_ptr())))

    def search_sharded_device(self, d_queries, out_tids, out_dist, out_count, strategy=1, nprobe=IVF_DEFAULT_NPROBE,
                              k=IVF_DEFAULT_K, max_candidates=0):
        """One batch on this rank's shard with the exchange done inside the library (ndbhip_ivf_search_sharded:
        probe all-gather, record all-gather, replay merge over the communicator of ndbhip_comm_init /
        ndbhip_comm_init_shm); every rank gets the full result.  Asynchronous on the library's stream over RCCL."""
        nq = d_queries.shape[0]
        check(lib().ndbhip_ivf_search_sharded(self._h, C.c_void_p(d_queries.data_ptr()), nq, strategy, nprobe, k,
                                              int(max_candidates), C.c_void_p(out_tids.data_ptr()),
                                              C.c_void_p(out_dist.data_ptr()), C.c_void_p(out_count.data_ptr())))

    def search_partial_device(self, d_queries, out_cand, out_ncand, out_total, strategy=1,
                              nprobe=IVF_DEFAULT_NPROBE, k=IVF_DEFAULT_K, max_candidates=0):
        nq = d_queries.shape[0]
        check(lib().ndbhip_ivf_search_partial_device(
            self._h, C.c_void_p(d_queries.data_ptr()), nq, strategy, nprobe, k, int(max_candidates),
            C.c_void_p(out_cand.data_ptr()), C.c_void_p(out_ncand.data_ptr()), C.c_void_p(out_total.data_ptr())))

    def search_partial_probes_device(self, d_queries, d_probes, out_cand, out_ncand, out_total, strategy=1,
                                     nprobe=IVF_DEFAULT_NPROBE, k=IVF_DEFAULT_K, max_candidates=0):
        """search_partial_device with the probes given ([nq, nprobe] int32 device tensor, contiguous)."""
        nq = d_queries.shape[0]
        assert d_probes.is_contiguous() and tuple(d_probes.shape) == (nq, nprobe)
        check(lib().ndbhip_ivf_search_partial_probes_device(
            self._h, C.c_void_p(d_queries.data_ptr()), nq, strategy, nprobe, k, int(max_candidates),
            C.c_void_p(d_probes.data_ptr()), C.c_void_p(out_cand.data_ptr()), C.c_void_p(out_ncand.data_ptr()),
            C.c_void_p(out_total.data_ptr())))

    def select_clusters_device(self, d_queries, d_out_probes, nprobe=IVF_DEFAULT_NPROBE):
        """ivfSelectClusters for device-resident queries -> d_out_probes [nq, nprobe] int32 (device)."""
        nq = d_queries.shape[0]
        assert d_out_probes.is_contiguous() and tuple(d_out_probes.shape) == (nq, nprobe)
        check(lib().ndbhip_ivf_select_clusters_device(self._h, C.c_void_p(d_queries.data_ptr()), nq, nprobe,
                                                      C.c_void_p(d_out_probes.data_ptr())))

    def select_clusters(self, queries, nprobe=IVF_DEFAULT_NPROBE):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        out = np.zeros((q.shape[0], nprobe), dtype=np.int32)
        check(lib().ndbhip_ivf_select_clusters(self._h, _ptr(q), q.shape[0], nprobe, _ptr(out)))
        return out


class IvfScan:
    """ivfbeginscan / ivfrescan / ivfgettuple / ivfendscan over an IvfIndex."""

    def __init__(self, index: IvfIndex, ref_compat: bool = False):
        # ivfbeginscan: ivf_am.c:1412-1437
        self.index = index
        self.ref_compat = ref_compat
        self.query = None
        self.strategy = 1
        self.nprobe = IVF_DEFAULT_NPROBE
        self.k = IVF_DEFAULT_K
        self.first_call = True
        self.result_count = 0
        self.results = None
        self.distances = None
        self.current = 0
        self.xs_heaptid = None
        self.xs_orderbyval = None

    def rescan(self, query, strategy=1, nprobe=None, k=None):
        # ivfrescan: ivf_am.c:1439-1545 — resets state, copies the query
        self.first_call = True
        self.result_count = 0
        self.current = 0
        self.results = self.distances = None
        self.query = None if query is None else np.array(query, dtype=np.float32, copy=True)
        if self.ref_compat:
            # Q1: every opclass registers strategy 1; Q4: nprobe pinned; Q3: k pinned
            self.strategy, self.nprobe, self.k = 1, IVF_DEFAULT_NPROBE, IVF_DEFAULT_K
        else:
            self.strategy = strategy
            self.nprobe = IVF_DEFAULT_NPROBE if nprobe is None else int(nprobe)
            self.k = IVF_DEFAULT_K if k is None else int(k)

    def gettuple(self) -> bool:
        # ivfgettuple: ivf_am.c:1911-2027
        if self.query is None:
            return False
        if self.first_call:
            if self.query.shape[0] != self.index.dim:       # :1961-1972
                self.first_call = False
                self.result_count = 0
                return False
            cap = self.k * 10 if self.ref_compat else 0     # :1743
            t, d, c = self.index.search(self.query[None, :], self.strategy, self.nprobe, self.k, cap)
            self.result_count = int(c[0])
            self.results, self.distances = t[0], d[0]
            self.first_call = False
            self.current = 0
        if self.current < self.result_count:                # :2011-2024
            self.xs_heaptid = self.results[self.current]
            self.xs_orderbyval = self.distances[self.current]
            self.current += 1
            return True
        return False

    def endscan(self):
        self.query = self.results = self.distances = None
