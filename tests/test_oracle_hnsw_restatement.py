"""A second restatement of hnswSearch (src/index/hnsw_am.c:1545-2080) and hnswComputeDistance (:1301-1345), in plain
Python from the reference source, against the C oracle's ndbo_hnsw_search on graphs the oracle itself builds
(hnswInsertNode's restatement) — greedy descent with `do { } while (foundBetter)` over the upper levels, then the
level-0 walk that stops as soon as efSearch candidates exist (quirk Q10: breadth-first until ef, not best-first),
replace-the-worst inside the node being expanded, swap-based selection sort of the first k.  Blocks, ranks, float4
bits and the number of distance evaluations must agree.  CPU only."""
import math

import numpy as np
import pytest

from oracle import ndbo

F = np.float32
INVALID = 0xFFFFFFFF
HNSW_MAX_LEVEL = 16


def ref_distance(a, b, strategy):
    """hnswComputeDistance: double accumulators; the DIFFERENCE and the PRODUCTS are float4 operations"""
    if strategy == 1:
        s = 0.0
        for x, y in zip(a, b):
            d = float(F(x - y))
            s += d * d
        return F(math.sqrt(s))
    dot = n1 = n2 = 0.0
    for x, y in zip(a, b):
        dot += float(F(x * y))
        if strategy == 2:
            n1 += float(F(x * x))
            n2 += float(F(y * y))
    if strategy == 3:
        return F(-dot)
    n1, n2 = math.sqrt(n1), math.sqrt(n2)
    if n1 == 0.0 or n2 == 0.0:
        return F(2.0)
    return F(1.0 - dot / (n1 * n2))                  # (double) 1.0f - double / double, then (float4)


def ref_hnsw_search(g, q, strategy, ef, k):
    vecs, levels, ncount, nbrs = g["vecs"], g["levels"], g["ncount"], g["nbrs"]
    nblocks, m = g["nblocks"], g["m"]
    evals = 0

    def valid_block(b):
        return b != INVALID and b < nblocks

    def clamp(n):                                    # hnswValidateNeighborCount
        return 0 if n < 0 else min(int(n), 2 * m)

    def dist(b):
        nonlocal evals
        evals += 1
        return ref_distance(q, vecs[b], strategy)

    if g["entry_point"] == INVALID:
        return [], [], 0
    current, level0 = g["entry_point"], g["entry_level"]
    if level0 < 0 or level0 >= HNSW_MAX_LEVEL:
        level0 = 0
    for level in range(level0, 0, -1):
        while True:                                  # do { ... } while (foundBetter)
            found = False
            if not valid_block(current):
                break
            lv = int(levels[current])
            if lv < 0 or lv >= HNSW_MAX_LEVEL:
                break
            cur_d = dist(current)
            if lv >= level:
                node = current                       # `neighbors` and its count stay those of the node being expanded
                for i in range(clamp(ncount[node, level])):
                    nb = int(nbrs[node, level, i])
                    if not valid_block(nb):
                        continue
                    nd = dist(nb)
                    if nd < cur_d:
                        current, cur_d, found = nb, nd, True
            if not found:
                break
    if not valid_block(current):
        return [], [], evals
    lv = int(levels[current])
    if lv < 0 or lv >= HNSW_MAX_LEVEL:
        return [], [], evals
    cand, cdist = [current], [dist(current)]
    visited = {current}
    i = 0
    while i < len(cand) and len(cand) < ef:          # for (i = 0; i < candidateCount && candidateCount < efSearch; i++)
        c = cand[i]
        i += 1
        if not valid_block(c):
            continue
        lv = int(levels[c])
        if lv < 0 or lv >= HNSW_MAX_LEVEL:
            continue
        for j in range(clamp(ncount[c, 0])):
            nb = int(nbrs[c, 0, j])
            if not valid_block(nb) or nb in visited:
                continue
            nd = dist(nb)
            visited.add(nb)
            if len(cand) < ef:
                cand.append(nb)
                cdist.append(nd)
            else:
                worst, wd = 0, cdist[0]
                for l in range(1, min(len(cand), ef)):
                    if cdist[l] > wd:
                        wd, worst = cdist[l], l
                if nd < wd:
                    cand[worst], cdist[worst] = nb, nd
    n = len(cand)
    idx = list(range(n))
    for i in range(min(k, n)):
        mi, md = i, cdist[idx[i]]
        for j in range(i + 1, n):
            if cdist[idx[j]] < md:
                md, mi = cdist[idx[j]], j
        if mi != i:
            idx[i], idx[mi] = idx[mi], idx[i]
    top = min(k, n)
    return [cand[idx[i]] for i in range(top)], [cdist[idx[i]] for i in range(top)], evals


@pytest.mark.parametrize("kind,strategy", [("normal", 1), ("normal", 2), ("normal", 3), ("integer", 1), ("unit", 2)])
def test_c_oracle_hnsw_search_equals_the_python_restatement(kind, strategy):
    rng = np.random.default_rng(17 * strategy + len(kind))
    for _ in range(3):
        dim = int(rng.choice([4, 9]))
        n = int(rng.integers(60, 220))
        m = int(rng.choice([4, 8]))
        if kind == "integer":
            base = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)          # ties: order of discovery decides
        else:
            base = rng.standard_normal((n, dim)).astype(np.float32)
        if kind == "unit":
            base /= np.linalg.norm(base, axis=1, keepdims=True) + 1e-9
            base[3] = 0                                                            # a zero vector: cosine 2.0f
        g = ndbo.HnswGraph(dim, m=m, ef_construction=20, cap_nodes=n + 2)
        r = rng.uniform(1e-12, 1.0, n)
        lv = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)     # hnsw_am.c:1143-1161
        for i in range(n):
            g.insert(base[i], i, int(lv[i]))
        arr = g.arrays()
        for _ in range(12):
            q = (rng.integers(-2, 3, dim).astype(np.float32) if kind == "integer"
                 else rng.standard_normal(dim).astype(np.float32))
            ef = int(rng.choice([1, 5, 16, 64]))
            k = int(rng.choice([1, 4, 10, 30]))
            blocks_p, dist_p, evals_p = ref_hnsw_search(arr, q, strategy, ef, k)
            blocks_c, dist_c, evals_c = g.search(q, strategy, ef, k)
            assert list(blocks_c) == blocks_p, (kind, strategy, ef, k)
            assert np.array_equal(dist_c.view(np.uint32), np.asarray(dist_p, np.float32).view(np.uint32)), (kind, strategy)
            assert evals_c == evals_p, (kind, strategy, ef, k, evals_c, evals_p)
