"""The `intended` search under the operator class's strategy (oracle/ndb_oracle_hnsw2.c "THE OPERATOR CLASS'S METRIC",
ndbo_h2_search_s) against a second, independent statement of the same definition in Python: the fp64 walk keys (L2: the squared distance; cosine: -dot * 1 / sqrt(|x|^2); inner product: -dot) by the fixed
64-partial tree (float4 rows: element i -> partial i mod 64; walk rows: (i / 4) mod 64), a best-first layer search over a
plain sorted list, the greedy descent, and the re-score of the result set with hnswComputeDistance's arithmetic
(src/index/hnsw_am.c:1301-1345, restated in tests/test_oracle_hnsw_restatement.py).  CPU only."""
import math

import numpy as np
import pytest

from oracle import ndbo
from tests.test_oracle_hnsw_restatement import ref_distance

INVALID = 0xFFFFFFFF


def _tree(terms, group4):
    """sum of float64 terms: partial p takes its elements in increasing order, then the butterfly 32 .. 1"""
    p = [0.0] * 64
    for i, t in enumerate(terms):
        j = ((i >> 2) & 63) if group4 else (i & 63)
        p[j] = p[j] + t
    off = 32
    while off:
        for i in range(off):
            p[i] = p[i] + p[i + off]
        off >>= 1
    return p[0]


def _half_image(x):
    """float4_to_fp16 then fp16_to_float (quantization.c:141-218), through the oracle's own walk-row encoder"""
    out = np.zeros(x.size, np.uint16)
    ndbo.lib().ndbo_h2_walk_rows(np.ascontiguousarray(x, np.float32).reshape(-1), x.size, out)
    return out.view(np.float16).astype(np.float32).reshape(x.shape)


def walk_key(q, x, strategy, group4):
    q64, x64 = q.astype(np.float64), x.astype(np.float64)
    if strategy == 1:
        d = (q - x).astype(np.float32).astype(np.float64)            # fl32(q_i - x_i), widened
        return _tree(list(d * d), group4)
    dot = _tree(list(q64 * x64), group4)
    if strategy == 3:
        return -dot
    nx = _tree(list(x64 * x64), group4)
    rinv = 1.0 / math.sqrt(nx) if nx > 0.0 else 0.0                  # a constant of the node (the device keeps it)
    return -dot * rinv


def search_py(a, rows, q, strategy, ef, k, group4):
    """a: the graph's arrays; rows: what the walk reads (the float4 rows, or their fp16 images)"""
    evals = 0
    if a["entry_point"] == INVALID or k < 1:
        return [], [], 0
    ef = max(ef, k)
    def key(b):
        nonlocal evals
        evals += 1
        return walk_key(q, rows[b], strategy, group4)

    def nbrs(b, level):
        cnt = int(a["ncount"][b, level]) if a["levels"][b] >= level else 0      # lists exist up to the node's own level
        return [int(e) for e in a["nbrs"][b, level, :cnt] if e != INVALID and e < a["nblocks"]]

    cur = int(a["entry_point"])
    curd = key(cur)
    for lc in range(int(a["entry_level"]), 0, -1):
        while True:
            best = (curd, cur)
            for e in nbrs(cur, lc):
                best = min(best, (key(e), e))
            if best[1] == cur:
                break
            curd, cur = best
    visited = {cur}
    res = [(curd, cur)]
    expanded = set()
    while True:
        cand = [r for r in res if r[1] not in expanded]
        if not cand:
            break
        c = min(cand)
        expanded.add(c[1])
        for e in nbrs(c[1], 0):
            if e in visited:
                continue
            visited.add(e)
            item = (key(e), e)
            if len(res) < ef:
                res.append(item)
            elif item < max(res):
                res[res.index(max(res))] = item
    res.sort()
    if strategy == 1 and not group4:
        out = res[:k]
        return [b for _, b in out], [np.float32(math.sqrt(d)) for d, _ in out], evals
    if strategy == 1:
        # walk rows: re-score on the float4 rows with the definition's d2, ascending again
        rs = sorted((walk_key(q, a["vecs"][b], 1, False), b) for _, b in res)
        evals += len(res)
        return [b for _, b in rs[:k]], [np.float32(math.sqrt(d)) for d, _ in rs[:k]], evals
    rs = sorted((float(ref_distance(q, a["vecs"][b], strategy)), b) for _, b in res)
    evals += len(res)
    return [b for _, b in rs[:k]], [np.float32(d) for d, _ in rs[:k]], evals


@pytest.mark.parametrize("kind,n,dim,m", [("scaled", 500, 24, 6), ("plain", 400, 70, 8), ("zeros", 300, 16, 4)])
def test_c_definition_equals_the_python_statement(kind, n, dim, m):
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    if kind == "scaled":
        base = (base * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
    if kind == "zeros":
        base[rng.integers(0, n, 12)] = 0.0
        q[1] = 0.0
    levels = np.clip((-np.log(rng.uniform(1e-12, 1.0, n)) * np.float32(0.36)).astype(np.int32), 0, 15)
    g = ndbo.HnswGraph(dim, m, 32, cap_nodes=n + 1)
    g.build_intended(base, levels, batch_div=8, batch_max=64, select=1)
    a = g.arrays()
    a["nbrs"] = a["nbrs"].reshape(a["nblocks"], 16, 2 * m)
    a["ncount"] = a["ncount"].reshape(a["nblocks"], 16)
    a["vecs"] = a["vecs"].reshape(a["nblocks"], dim)
    w16 = g.walk_rows() if dim % 4 == 0 else None
    half = _half_image(a["vecs"]) if w16 is not None else None
    orders = {}
    for strategy in (1, 2, 3):
        for ef, k in ((24, 5), (3, 3)):
            for walk in ([False, True] if w16 is not None else [False]):
                for i in range(len(q)):
                    eb, ed, ns = g.search_intended_s(q[i], strategy, ef, k, w16=w16 if walk else None)
                    pb, pd, pn = search_py(a, half if walk else a["vecs"], q[i], strategy, ef, k, walk)
                    assert list(eb) == pb and ns == pn, (strategy, ef, walk, i, list(eb), pb, ns, pn)
                    assert np.array_equal(np.asarray(pd, np.float32).view(np.uint32), ed.view(np.uint32))
                    if ef == 24 and not walk:
                        orders[(strategy, i)] = pb
    # strategy 1 through the new entry point is ndbo_h2_search / _w16
    for i in range(len(q)):
        e1 = g.search_intended(q[i], 24, 5)
        e2 = g.search_intended_s(q[i], 1, 24, 5)
        assert np.array_equal(e1[0], e2[0]) and np.array_equal(e1[1].view(np.uint32), e2[1].view(np.uint32)) and e1[2] == e2[2]
    if kind == "scaled":
        assert any(orders[(1, i)] != orders[(2, i)] for i in range(len(q))) and any(orders[(2, i)] != orders[(3, i)] for i in range(len(q)))
    with pytest.raises(ValueError):
        g.search_intended_s(q[0], 4, 24, 5)
