"""A second, independent restatement of the reference's IVF scan — written in plain Python straight from
src/index/ivf_am.c (ivfComputeDistance :1550-1592, ivfSelectClusters :1597-1717, ivfCollectCandidates :1722-1909,
and the palloc0'd cluster array of ivfgettuple :1978) — against the C oracle (oracle/ndb_oracle.c), which every GPU
parity test is measured with.  Two restatements by different routes agreeing on the quirks (FLT_MAX start values,
strict <, the swap-based selection sort, list 0 read again for the probe slots beyond nlists, the k*10 candidate
cap, dead and foreign-dim entries) is what can be had without a PostgreSQL to run the reference itself.  CPU only."""
import numpy as np
import pytest

from oracle import ndbo

F = np.float32
FLT_MAX = F(np.finfo(np.float32).max)


def ref_distance(a, b, strategy):
    """ivfComputeDistance: float4 accumulators, one rounded operation after the other"""
    with np.errstate(all="ignore"):
        if strategy == 2:
            dot, n1, n2 = F(0), F(0), F(0)
            for x, y in zip(a, b):
                dot = F(dot + F(x * y))
                n1 = F(n1 + F(x * x))
                n2 = F(n2 + F(y * y))
            n1, n2 = np.sqrt(n1), np.sqrt(n2)
            if n1 == 0 or n2 == 0:
                return F(1)
            return F(F(1) - F(dot / F(n1 * n2)))
        s = F(0)                                     # case 1 and default
        for x, y in zip(a, b):
            d = F(x - y)
            s = F(s + F(d * d))
        return np.sqrt(s)


def ref_select_clusters(cent, cent_dim, nlists_meta, q, nprobe, selected):
    """ivfSelectClusters; `selected` is the caller's palloc0'd array of the ORIGINAL nprobe entries"""
    dim = len(q)
    maxoff = len(cent)
    if maxoff == 0:
        selected[:nprobe] = -1
        return
    nlists = nlists_meta
    if nprobe > nlists:
        nprobe = nlists
    if nlists > maxoff:
        nlists = maxoff
    if nprobe > nlists:
        nprobe = nlists
    dist = [FLT_MAX] * nlists
    for i in range(min(nlists, maxoff)):
        if cent_dim is not None and cent_dim[i] != dim:
            dist[i] = FLT_MAX
            continue
        dist[i] = ref_distance(q, cent[i], 1)        # "Use L2 for cluster selection"
    for i in range(nprobe):
        best_idx, best = -1, FLT_MAX
        for j in range(nlists):
            if j in selected[:i]:
                continue
            if dist[j] < best:                       # NaN and anything >= FLT_MAX never wins
                best, best_idx = dist[j], j
        selected[i] = best_idx


def ref_search(img, q, strategy, nprobe, k, cap):
    """ivfgettuple's first call: palloc0(nprobe) clusters, select, collect, selection sort, top-k"""
    selected = np.zeros(nprobe, np.int64)            # palloc0: the slots ivfSelectClusters leaves alone stay 0
    ref_select_clusters(img.centroids, img.centroid_dim, img.nlists, q, nprobe, selected)
    maxoff = len(img.centroids)
    max_cand = cap if cap > 0 else 10 ** 9
    cand_tid, cand_d = [], []
    for i in range(nprobe):
        if len(cand_d) >= max_cand:
            break
        c = int(selected[i])
        if c < 0 or c >= maxoff:
            continue
        for r in range(int(img.list_off[c]), int(img.list_off[c + 1])):
            if len(cand_d) >= max_cand:
                break
            if img.live is not None and not img.live[r]:
                continue
            cand_d.append(ref_distance(q, img.vecs[r], strategy))
            cand_tid.append(r)
    n = len(cand_d)
    if n == 0:
        return [], []
    idx = list(range(n))
    actual_k = min(k, n)
    for i in range(actual_k):
        best_idx, best = i, cand_d[idx[i]]
        for j in range(i + 1, n):
            if cand_d[idx[j]] < best:
                best, best_idx = cand_d[idx[j]], j
        if best_idx != i:
            idx[i], idx[best_idx] = idx[best_idx], idx[i]
    return [cand_tid[idx[i]] for i in range(actual_k)], [cand_d[idx[i]] for i in range(actual_k)]


def make_image(rng, n, dim, nlists, kind):
    if kind == "integer":                            # many exact ties: the selection sort's swaps show
        base = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
    elif kind == "huge":                             # sums beyond FLT_MAX: +inf distances, never < FLT_MAX
        base = (rng.standard_normal((n, dim)) * 1e19).astype(np.float32)
        base[::3] = rng.standard_normal((len(base[::3]), dim)).astype(np.float32)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    if kind == "nan":
        base[rng.integers(0, n, 3), rng.integers(0, dim, 3)] = np.nan
    base[rng.integers(0, n, n // 8)] = base[rng.integers(0, n, n // 8)]       # duplicates
    cent = base[rng.choice(n, nlists, replace=False)].copy()
    asg = rng.integers(0, nlists, n)
    order = np.argsort(asg, kind="stable")
    off = np.zeros(nlists + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(asg, minlength=nlists))
    live = None
    if rng.random() < 0.5:
        live = (rng.random(n) > 0.1).astype(np.uint8)
    cdim = None
    if rng.random() < 0.3 and nlists > 2:
        cdim = np.full(nlists, dim, np.int32)
        cdim[rng.integers(0, nlists)] = dim + 1      # a centroid of another dimension: distance stays FLT_MAX
    return ndbo.IvfImage(cent, off, base[order], ndbo.tids_from_rows(order), live=live, centroid_dim=cdim), order


@pytest.mark.parametrize("kind", ["normal", "integer", "huge", "nan"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_c_oracle_equals_the_python_restatement_of_the_reference(kind, seed):
    rng = np.random.default_rng(100 * seed + len(kind))
    for _ in range(6):
        dim = int(rng.choice([3, 8, 17]))
        n = int(rng.integers(30, 160))
        nlists = int(rng.integers(1, 9))
        img, _ = make_image(rng, n, dim, nlists, kind)
        for _ in range(4):
            q = rng.integers(-2, 3, dim).astype(np.float32) if kind == "integer" else rng.standard_normal(dim).astype(np.float32)
            if kind == "huge" and rng.random() < 0.5:
                q = (q * 1e19).astype(np.float32)
            strategy = int(rng.choice([1, 1, 2]))
            nprobe = int(rng.integers(1, nlists + 4))            # up to 3 probe slots beyond nlists: list 0 again
            k = int(rng.choice([1, 5, 10, 40]))
            cap = int(rng.choice([0, k * 10, 7]))
            # the cluster choice alone
            sel_c = img.select_clusters(q, nprobe)
            sel_p = np.zeros(nprobe, np.int64)
            ref_select_clusters(img.centroids, img.centroid_dim, img.nlists, q, nprobe, sel_p)
            eff = min(nprobe, img.nlists, len(img.centroids))
            assert np.array_equal(sel_c[:eff], sel_p[:eff]), (kind, sel_c, sel_p)
            # the whole scan
            with np.errstate(all="ignore"):
                rows_p, dist_p = ref_search(img, q, strategy, nprobe, k, cap)
            t, d, _ = img.search(q, strategy, nprobe, k, cap)
            assert len(t) == len(rows_p), (kind, len(t), len(rows_p))
            exp_t = ndbo.tids_to_u64(img.tids[np.asarray(rows_p, np.int64)]) if rows_p else np.zeros(0, np.uint64)
            assert np.array_equal(ndbo.tids_to_u64(t), exp_t), (kind, strategy, nprobe, k, cap)
            assert np.array_equal(d.view(np.uint32), np.asarray(dist_p, np.float32).view(np.uint32)), (kind, strategy)


def test_the_case_round_6_fuzz_found_is_pinned_by_both_restatements():
    """tests/golden/fuzz_r6_dup_probe_seed.npz (the mirror and queries of the one mismatch round 6's long fuzz campaigns
    found: nlists = 1, nprobe = 2 — list 0 scanned twice, ivf_am.c:1978 — candidate cap 500 over 2 x 373 rows, k = 1): the C
    oracle and the Python restatement agree on it, every query has its neighbour, and the second pass of list 0 contributes
    exactly cap - 373 = 127 candidates (what the device's seed kernel has to count rows against, tests/test_gpu_screen16.py)."""
    import os
    from tests.util import oracle_image
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz_r6_dup_probe_seed.npz"))
    a = dict(centroids=z["centroids"], list_len=z["list_len"], rows=z["rows"], tids=z["tids"])
    img = oracle_image(a)
    q, k, nprobe, cap, strategy = z["q"], int(z["k"]), int(z["nprobe"]), int(z["cap"]), int(z["strategy"])
    assert img.nlists == 1 and nprobe == 2 and cap == 500 and len(a["rows"]) == 373 and k == 1 and strategy == 1
    for i in range(len(q)):
        rows_p, dist_p = ref_search(img, q[i], strategy, nprobe, k, cap)
        t, d, total = img.search(q[i], strategy, nprobe, k, cap)
        assert len(t) == len(rows_p) == 1
        assert np.array_equal(ndbo.tids_to_u64(t), ndbo.tids_to_u64(img.tids[np.asarray(rows_p, np.int64)]))
        assert np.array_equal(d.view(np.uint32), np.asarray(dist_p, np.float32).view(np.uint32))
        assert total == cap                                        # 373 rows of the first pass + 127 of the second
        # the neighbour is the float64 nearest row (no tie anywhere near on this data)
        d2 = ((a["rows"].astype(np.float64) - q[i].astype(np.float64)) ** 2).sum(1)
        assert rows_p[0] == int(np.argmin(d2))


# ---------------------------------------------------------------------------------------------------------------
# The build: kmeans_init / kmeans_run and their helpers (ivf_am.c:2070-2294) and the insert-time choice of a list
# (:812-814, :905-935), restated the same way.
# ---------------------------------------------------------------------------------------------------------------

def ref_l2_squared(a, b):
    """vector_distance_l2 (:2255-2269): the SQUARED distance, float4 accumulator"""
    s = F(0)
    for x, y in zip(a, b):
        d = F(x - y)
        s = F(s + F(d * d))
    return s


def ref_kmeans(data, k, max_iter=50, threshold=0.001):
    """kmeans_init + kmeans_run; returns centroids, assignments, counts, Lloyd iterations run, last cost"""
    n, dim = data.shape
    cent = np.zeros((k, dim), np.float32)
    for i in range(k):                               # "Initialize with random data points (KMeans++)": the first k
        if i < n:
            cent[i] = data[i]
    asg = np.zeros(n, np.int64)
    counts = np.zeros(k, np.int64)
    prev = FLT_MAX
    thr = F(threshold)                               # float4 threshold = IVF_CONVERGENCE_THRESHOLD
    iters = 0
    cost = F(0)
    with np.errstate(all="ignore"):
        for it in range(max_iter):
            counts[:] = 0                            # kmeans_assign
            for i in range(n):
                best, best_d = 0, FLT_MAX            # find_nearest_centroid: best = 0, bestDist = FLT_MAX
                for c in range(k):
                    d = ref_l2_squared(data[i], cent[c])
                    if d < best_d:
                        best_d, best = d, c
                asg[i] = best
                counts[best] += 1
            cent[:] = 0                              # kmeans_update_centroids: sums in sample order, then / count
            for i in range(n):
                c = asg[i]
                for j in range(dim):
                    cent[c, j] = F(cent[c, j] + data[i, j])
            for c in range(k):
                if counts[c] > 0:
                    for j in range(dim):
                        cent[c, j] = F(cent[c, j] / F(counts[c]))
            cost = F(0)                              # kmeans_compute_cost: float4 sum in sample order
            for i in range(n):
                cost = F(cost + ref_l2_squared(data[i], cent[asg[i]]))
            iters = it + 1
            if abs(float(F(prev - cost))) < float(thr):          # fabs(prevCost - cost) < state->threshold
                break
            prev = cost
    return cent, asg, counts, iters, cost


def ref_insert_list(cent, x):
    """ivfinsert's choice: min_idx = 0, min_dist = FLT_MAX, sqrtf of the float4 sum, strict <"""
    best, best_d = 0, FLT_MAX
    with np.errstate(all="ignore"):
        for i in range(len(cent)):
            d = np.sqrt(ref_l2_squared(x, cent[i]))
            if d < best_d:
                best_d, best = d, i
    return best


@pytest.mark.parametrize("kind", ["normal", "integer", "clustered", "huge"])
def test_c_oracle_kmeans_and_insert_rule_equal_the_python_restatement(kind):
    rng = np.random.default_rng(len(kind) * 7)
    for _ in range(5):
        dim = int(rng.choice([2, 5, 12]))
        k = int(rng.integers(1, 7))
        n = int(rng.integers(k, 70))
        if kind == "integer":                        # ties everywhere, duplicate initial centroids, empty clusters
            data = rng.integers(-1, 2, size=(n, dim)).astype(np.float32)
        elif kind == "clustered":
            cen = rng.standard_normal((3, dim)).astype(np.float32) * 5
            data = (cen[rng.integers(0, 3, n)] + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
        elif kind == "huge":                         # squared distances overflow: nothing is < FLT_MAX, everything goes to 0
            data = (rng.standard_normal((n, dim)) * 2e19).astype(np.float32)
        else:
            data = rng.standard_normal((n, dim)).astype(np.float32)
        cent_p, asg_p, cnt_p, it_p, cost_p = ref_kmeans(data, k)
        cent_c, asg_c, cnt_c, it_c, cost_c = ndbo.kmeans(data, k)
        assert it_c == it_p, (kind, it_c, it_p)
        assert np.array_equal(asg_c, asg_p) and np.array_equal(cnt_c, cnt_p), kind
        assert np.array_equal(cent_c.view(np.uint32), cent_p.view(np.uint32)), kind
        assert np.float32(cost_c).view(np.uint32) == np.float32(cost_p).view(np.uint32), (kind, cost_c, cost_p)
        # every row to the list ivfinsert would choose
        got = ndbo.ivf_assign_all(cent_c, data)
        exp = np.asarray([ref_insert_list(cent_p, data[i]) for i in range(n)])
        assert np.array_equal(got, exp), kind
