#!/usr/bin/env python3
"""Randomised parity campaign for the `intended` HNSW (csrc/ndbhip_hnsw2.h) against its sequential definition
(oracle/ndb_oracle_hnsw2.c): random sizes, dimensions, m, ef_construction, batch schedules, selection rules and data kinds
(unit, scaled, integer ties, offset, zero rows, duplicates); the device-built graph must equal the oracle's slot for slot, a
second batch of rows appended to it too, and the search — strategies 1 / 2 / 3, float4 and fp16 walk rows, random ef and k —
must return the oracle's blocks, float4 distance bits and evaluation counts.
usage: python tools/fuzz_hnsw.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import ndbo


def levels_of(rng, n):
    r = rng.uniform(1e-12, 1.0, n)
    return np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)


def one_case(rng, HnswIndex, _lib):
    dim = int(rng.choice([4, 20, 33, 48, 64, 100, 128, 256, 768, 1100]))
    n = int(rng.integers(50, 2500 if dim <= 256 else 900))
    m = int(rng.choice([4, 6, 8, 16, 24]))
    efc = int(rng.choice([8, 24, 40, 64, 100, 200]))
    bdiv, bmax = [(64, 1024), (16, 256), (8, 64), (1, 1), (32, 512), (4, 4096)][int(rng.integers(0, 6))]
    select = int(rng.choice([0, 1, 1, 2, 3, 5, 7]))
    nq = int(rng.choice([1, 7, 33, 64]))
    kind = str(rng.choice(["normal", "clustered", "integer", "offset", "scaled"]))
    if kind == "clustered":
        cen = rng.standard_normal((16, dim)).astype(np.float32)
        base = (cen[rng.integers(0, 16, n)] + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
        q = (cen[rng.integers(0, 16, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    elif kind == "integer":
        base = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
        q = rng.integers(-2, 3, size=(nq, dim)).astype(np.float32)
    elif kind == "offset":
        base = (100.0 + 0.1 * rng.standard_normal((n, dim))).astype(np.float32)
        q = (100.0 + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    elif kind == "scaled":
        base = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
        q = (rng.standard_normal((nq, dim)) * rng.uniform(0.2, 5.0, (nq, 1))).astype(np.float32)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
    if rng.random() < 0.3:
        base[rng.integers(0, n, max(1, n // 20))] = 0.0          # zero rows: cosine's 2.0 branch, 1 / |row| = 0 on the walk
    if rng.random() < 0.3:
        base[rng.integers(0, n, n // 10)] = base[rng.integers(0, n, n // 10)]      # duplicates: ties by block number
    if rng.random() < 0.2:
        q[0] = 0.0
    if rng.random() < 0.2:
        q[nq - 1] = base[int(rng.integers(0, n))]
    levels = levels_of(rng, n)
    n0 = n if rng.random() < 0.6 else int(rng.integers(1, n))    # rows of the first build; the rest is appended
    case = dict(dim=dim, n=n, n0=n0, m=m, efc=efc, bdiv=bdiv, bmax=bmax, select=select, nq=nq, kind=kind)
    if os.environ.get("FUZZ_TRACE"):
        print("CASE", case, flush=True)
    og = ndbo.HnswGraph(dim, m, efc, cap_nodes=n + 1)
    og.build_intended(base[:n0], levels[:n0], batch_div=bdiv, batch_max=bmax, select=select)
    _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(select))
    ix = HnswIndex(dim, m)
    try:
        ix.build_intended(base[:n0], ndbo.tids_from_rows(np.arange(n0)), levels[:n0], efc, batch_div=bdiv, batch_max=bmax)
        if n0 < n:
            og.build_intended(base[n0:], levels[n0:], tids=ndbo.tids_from_rows(np.arange(n0, n)), batch_div=bdiv, batch_max=bmax, select=select)
            ix.build_intended(base[n0:], ndbo.tids_from_rows(np.arange(n0, n)), levels[n0:], efc, batch_div=bdiv, batch_max=bmax, append=True)
        e, d = og.arrays(), ix.export()
        assert d["nblocks"] == n + 1 and d["entry_point"] == e["entry_point"] and d["entry_level"] == e["entry_level"], ("entry", case)
        assert np.array_equal(d["levels"][1:], e["levels"][1:]), ("levels", case)
        assert np.array_equal(d["ncount"][1:], e["ncount"][1:]), ("ncount", case, np.argwhere(d["ncount"] != e["ncount"])[:5])
        assert np.array_equal(d["nbrs"][1:], e["nbrs"][1:]), ("nbrs", case, np.argwhere(d["nbrs"] != e["nbrs"])[:5])
        w16 = og.walk_rows() if dim % 4 == 0 and dim <= 1024 else None
        for _ in range(3):
            strategy = int(rng.choice([1, 2, 3]))
            ef = int(rng.choice([1, 2, 8, 16, 63, 64, 65, 100, 200]))
            k = int(rng.choice([1, 2, 10, 16, 37]))
            walk16 = w16 is not None and rng.random() < 0.5
            ob, od, oc, oe = ix.search_intended(q, ef, k, walk16=walk16, strategy=strategy)
            for i in range(nq):
                eb, ed, ns = og.search_intended_s(q[i], strategy, ef, k, w16=w16 if walk16 else None)
                assert oc[i] == len(eb) and np.array_equal(ob[i, :oc[i]], eb), ("blocks", case, strategy, ef, k, walk16, i, ob[i], eb)
                assert np.array_equal(od[i, :oc[i]].view(np.uint32), ed.view(np.uint32)), ("bits", case, strategy, ef, k, walk16, i)
                assert oe[i] == ns, ("evaluations", case, strategy, ef, k, walk16, i, oe[i], ns)
    finally:
        _lib.check(_lib.lib().ndbhip_hnsw_set_intended_select(1))
        ix.close()
    return kind + ("/appended" if n0 < n else "")


def main():
    from neurondb_amd import HnswIndex, _lib
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    rng = np.random.default_rng(seed)
    t0, n, kinds = time.time(), 0, {}
    while time.time() - t0 < secs:
        kd = one_case(rng, HnswIndex, _lib)
        kinds[kd] = kinds.get(kd, 0) + 1
        n += 1
    print(f"fuzz_hnsw: {n} random graphs (build, append, 3 searches each) identical to the oracle (seed {seed}): {kinds}")


if __name__ == "__main__":
    main()
