"""oracle/ndbo_hnsw_search_layer (restatement of src/scan/hnsw_scan.c) against a second,
independent restatement written in plain Python from the same reference lines — the reference has no test,
fixture or caller for this file, so nothing of its own pins it (parity unpinned)."""
import numpy as np

from oracle import ndbo

INVALID = 0xFFFFFFFF
FLT_MAX = np.finfo(np.float32).max


def l2(a, b):
    """compute_l2_distance, hnsw_scan.c:105-118: fp32 sequential, sqrtf"""
    s = np.float32(0.0)
    for x, y in zip(a, b):
        d = np.float32(x) - np.float32(y)
        s = np.float32(s + np.float32(d * d))
    return np.float32(np.sqrt(s))


def search_layer_py(a, q, ef, k):
    vecs, ncount, nbrs, nb_total, m = a["vecs"], a["ncount"], a["nbrs"], a["nblocks"], a["m"]
    scored = 0
    entry, level = a["entry_point"], a["entry_level"]
    if entry == INVALID or level < 0:
        return [], [], 0
    readable = lambda b: b < nb_total and b != 0
    clamp = lambda c: max(0, min(int(c), 2 * m))
    while level > 0:                                        # :448-457 + hnswSearchLayerGreedy :485-636
        best, changed = entry, True
        while changed:
            changed = False
            if not readable(best):
                break
            nlist, nc = nbrs[best, level], clamp(ncount[best, level])
            bd = l2(q, vecs[best]); scored += 1
            for i in range(nc):
                n = int(nlist[i])
                if n == INVALID or not readable(n):
                    continue
                d = l2(q, vecs[n]); scored += 1
                if d < bd:
                    best, bd, changed = n, d, True
        entry, level = best, level - 1
    heap, cap = [], 2 * ef                                  # hnswSearchLayer0 :645-844

    def push(b, d):                                         # :235-266
        if len(heap) >= cap:
            return
        heap.append((b, d))
        i = len(heap) - 1
        while i > 0:
            p = (i - 1) // 2
            if heap[i][1] >= heap[p][1]:
                break
            heap[i], heap[p] = heap[p], heap[i]
            i = p

    def pop():                                              # :271-327
        top = heap[0]
        last = heap.pop()
        if heap:
            heap[0] = last
            i = 0
            while True:
                s, l, r = i, 2 * i + 1, 2 * i + 2
                if l < len(heap) and heap[l][1] < heap[s][1]:
                    s = l
                if r < len(heap) and heap[r][1] < heap[s][1]:
                    s = r
                if s == i:
                    break
                heap[i], heap[s] = heap[s], heap[i]
                i = s
        return top

    visited, res = {entry}, []
    push(entry, np.float32(0.0))
    while heap:
        block, dist = pop()
        if len(res) >= k and dist > res[k - 1][1]:
            continue
        if not readable(block):
            continue
        nlist, nc = nbrs[block, 0], clamp(ncount[block, 0])
        dist = l2(q, vecs[block]); scored += 1
        furthest = res[k - 1][1] if len(res) >= k else FLT_MAX
        for j in range(nc):
            n = int(nlist[j])
            if n == INVALID or not readable(n) or n in visited:
                continue
            d = l2(q, vecs[n]); scored += 1
            if d < furthest or len(res) < k:
                push(n, d)
                visited.add(n)
        if len(res) < k:                                    # hnswAddResult :333-365
            res.append((block, dist))
        else:
            wi, wd = 0, res[0][1]
            for i in range(1, len(res)):
                if res[i][1] > wd:
                    wi, wd = i, res[i][1]
            if dist < wd:
                res[wi] = (block, dist)
    return [b for b, _ in res], [d for _, d in res], scored


def _graph(n, dim, m, efc, seed, integer=False):
    rng = np.random.default_rng(seed)
    vecs = (rng.integers(-2, 3, size=(n, dim)) if integer else rng.standard_normal((n, dim))).astype(np.float32)
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 4)
    L = ndbo.lib()
    for i in range(n):
        g.insert(vecs[i], i, L.ndbo_hnsw_level_from_uniform(float(rng.uniform(1e-9, 1.0)), np.float32(0.36)))
    return g, vecs, rng


def test_c_restatement_equals_the_python_restatement():
    for n, dim, m, efc, integer in ((400, 8, 4, 16, False), (300, 12, 6, 24, True), (1, 4, 4, 8, False)):
        g, vecs, rng = _graph(n, dim, m, efc, seed=n + dim, integer=integer)
        a = g.arrays()
        qs = (rng.integers(-2, 3, size=(6, dim)) if integer else rng.standard_normal((6, dim))).astype(np.float32)
        qs[0] = vecs[0]
        for ef, k in ((32, 10), (1, 5), (2, 1), (8, 40)):
            for q in qs:
                b, d, ns = g.search_layer(q, ef, k)
                eb, ed, es = search_layer_py(a, q, ef, k)
                assert b.tolist() == eb and ns == es
                assert np.array_equal(d.view(np.uint32), np.array(ed, np.float32).view(np.uint32))


def test_search_layer_properties():
    g, vecs, rng = _graph(600, 8, 6, 24, seed=5)
    empty = ndbo.HnswGraph(8, m=6, ef_construction=24, cap_nodes=4)
    assert len(empty.search_layer(vecs[0], 16, 5)[0]) == 0          # no entry point: no rows (:396-402)
    for q in rng.standard_normal((10, 8)).astype(np.float32):
        b, d, ns = g.search_layer(q, 32, 10)
        assert 1 <= len(b) <= 10 and len(set(b.tolist())) == len(b) and ns >= len(b)
        # every distance is compute_l2_distance of the returned node
        assert np.array_equal(d.view(np.uint32), np.array([l2(q, vecs[x - 1]) for x in b]).view(np.uint32))
