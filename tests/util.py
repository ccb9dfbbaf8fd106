"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import ndbo


def make_ivf_arrays(n, dim, nlists, seed, dup_frac=0.0, zero_rows=0, empty_lists=(), integer=False):
    """Random IVF image (any centroids/lists are a valid index: parity must hold for all).
    Returns dict(centroids, list_len, rows (list-major), tids, base_order)."""
    rng = np.random.default_rng(seed)
    if integer:
        base = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)   # many exact ties
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    if dup_frac > 0 and n > 4:
        nd = int(n * dup_frac)
        src = rng.integers(0, n, nd)
        dst = rng.integers(0, n, nd)
        base[dst] = base[src]
    if zero_rows:
        base[rng.integers(0, n, zero_rows)] = 0.0
    cent = base[rng.choice(n, size=nlists, replace=(n < nlists))].copy()
    cent += (0.05 * rng.standard_normal(cent.shape)).astype(np.float32)
    d2 = ((base[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1) \
        if n * nlists * dim < 5e7 else None
    if d2 is None:
        bb = (base.astype(np.float64) ** 2).sum(1)[:, None]
        cc = (cent.astype(np.float64) ** 2).sum(1)[None]
        d2 = bb + cc - 2.0 * base.astype(np.float64) @ cent.astype(np.float64).T
    for e in empty_lists:
        d2[:, e] = np.inf
    asg = d2.argmin(1)
    order = np.argsort(asg, kind="stable")
    list_len = np.bincount(asg, minlength=nlists).astype(np.int64)
    return dict(centroids=cent, list_len=list_len, rows=np.ascontiguousarray(base[order]),
                tids=ndbo.tids_from_rows(order), order=order, base=base)


def oracle_image(a, nlists=None):
    off = np.zeros(len(a["list_len"]) + 1, dtype=np.int64)
    off[1:] = np.cumsum(a["list_len"])
    return ndbo.IvfImage(a["centroids"], off, a["rows"], a["tids"], nlists=nlists)


def oracle_search_batch(img, queries, strategy, nprobe, k, cap=0):
    nq = len(queries)
    T = np.zeros((nq, k), dtype=np.uint64)
    D = np.zeros((nq, k), dtype=np.float32)
    Cn = np.zeros(nq, dtype=np.int32)
    scored = 0
    for i, q in enumerate(queries):
        t, d, ns = img.search(q, strategy, nprobe, k, cap)
        Cn[i] = len(t)
        T[i, :len(t)] = ndbo.tids_to_u64(t)
        D[i, :len(t)] = d
        scored += ns
    return T, D, Cn, scored


def assert_same_results(got_t, got_d, got_c, exp_t, exp_d, exp_c):
    """ids and ranks identical, distances bit-identical."""
    assert np.array_equal(got_c, exp_c), (got_c, exp_c)
    for i in range(len(exp_c)):
        c = exp_c[i]
        assert np.array_equal(ndbo.tids_to_u64(got_t[i, :c]), exp_t[i, :c]), \
            (i, ndbo.tids_to_u64(got_t[i, :c]), exp_t[i, :c], got_d[i, :c], exp_d[i, :c])
        assert np.array_equal(got_d[i, :c].view(np.uint32), exp_d[i, :c].view(np.uint32)), \
            (i, got_d[i, :c], exp_d[i, :c])
