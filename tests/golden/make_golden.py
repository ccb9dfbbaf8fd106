#!/usr/bin/env python3
"""Generates tests/golden/ivf_small.npz and hnsw_small.npz: small seeded inputs plus the outputs of the
CPU oracle (oracle/ndb_oracle.c) on them.  The reference itself cannot be run here (PostgreSQL absent, see
DESIGN.md "Oracle"), so these freeze the oracle's behaviour: a later change to the oracle or to the HIP path
that alters any id, rank, distance bit or k-means centroid shows up against the committed file.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ndbo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ivf():
    rng = np.random.default_rng(20251205)
    n, dim, nlists = 1500, 32, 12
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[700:720] = base[100:120]                      # duplicates -> distance ties
    base[5] = 0.0                                      # zero vector (cosine zero-norm rule)
    img, asg, iters = ndbo.build_ivf_image(base, nlists, max_iter=50)
    q = rng.standard_normal((16, dim)).astype(np.float32)
    q[0] = base[100]
    q[1] = 0.0
    out = {"base": base, "queries": q, "nlists": nlists, "kmeans_iters": iters,
           "centroids": img.centroids, "list_off": img.list_off, "assign": asg}
    for name, (strategy, nprobe, k, cap) in {"l2": (1, 4, 10, 0), "cos": (2, 4, 10, 0), "ip": (3, 4, 10, 0),
                                             "refcompat": (1, 10, 10, 100), "k37": (1, 6, 37, 0)}.items():
        T = np.zeros((len(q), k), np.uint64)
        D = np.zeros((len(q), k), np.float32)
        C = np.zeros(len(q), np.int32)
        for i, qq in enumerate(q):
            t, d, _ = img.search(qq, strategy, nprobe, k, cap)
            C[i] = len(t)
            T[i, :len(t)] = ndbo.tids_to_u64(t)
            D[i, :len(t)] = d
        out[f"{name}_tids"], out[f"{name}_dist"], out[f"{name}_count"] = T, D, C
        out[f"{name}_params"] = np.array([strategy, nprobe, k, cap])
    np.savez_compressed(os.path.join(HERE, "ivf_small.npz"), **out)


def hnsw():
    rng = np.random.default_rng(777)
    n, dim, m, efc = 400, 24, 6, 24
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    levels = np.array([ndbo.lib().ndbo_hnsw_level_from_uniform(float(r), np.float32(0.36))
                       for r in rng.uniform(1e-9, 1.0, n)], np.int32)
    g = ndbo.HnswGraph(dim, m=m, ef_construction=efc, cap_nodes=n + 2)
    for i in range(n):
        g.insert(vecs[i], i, levels[i])
    a = g.arrays()
    q = rng.standard_normal((10, dim)).astype(np.float32)
    q[0] = vecs[17]
    out = {"vecs": vecs, "levels_in": levels, "m": m, "efc": efc, "queries": q,
           "g_levels": a["levels"], "g_ncount": a["ncount"], "g_nbrs": a["nbrs"], "g_tids": a["tids"],
           "entry_point": a["entry_point"], "entry_level": a["entry_level"]}
    for strategy in (1, 2, 3):
        B = np.zeros((len(q), 10), np.uint32)
        D = np.zeros((len(q), 10), np.float32)
        C = np.zeros(len(q), np.int32)
        S = np.zeros(len(q), np.int64)
        for i, qq in enumerate(q):
            b, d, ns = g.search(qq, strategy, 32, 10)
            C[i], S[i] = len(b), ns
            B[i, :len(b)], D[i, :len(b)] = b, d
        out[f"s{strategy}_blocks"], out[f"s{strategy}_dist"] = B, D
        out[f"s{strategy}_count"], out[f"s{strategy}_scored"] = C, S
    # src/scan/hnsw_scan.c (hnsw_search_layer) on the same graph: slot order, compute_l2_distance
    for ef, k in ((32, 10), (2, 5)):
        B = np.zeros((len(q), k), np.uint32)
        D = np.zeros((len(q), k), np.float32)
        C = np.zeros(len(q), np.int32)
        S = np.zeros(len(q), np.int64)
        for i, qq in enumerate(q):
            b, d, ns = g.search_layer(qq, ef, k)
            C[i], S[i] = len(b), ns
            B[i, :len(b)], D[i, :len(b)] = b, d
        out[f"layer{ef}_blocks"], out[f"layer{ef}_dist"] = B, D
        out[f"layer{ef}_count"], out[f"layer{ef}_scored"] = C, S
    np.savez_compressed(os.path.join(HERE, "hnsw_small.npz"), **out)


if __name__ == "__main__":
    ivf()
    hnsw()
    print("wrote", os.listdir(HERE))
