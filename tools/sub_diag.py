"""How much of the probed rows does the sublist exclusion keep? (diagnostic for the overflow test's table)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from neurondb_amd import _lib as lib, IvfIndex
from oracle import ndbo
lib.ensure_init()
dim, nlists, nprobe, nq, k = int(os.environ.get("DIM", 32)), 40, int(os.environ.get("NPROBE", 40)), int(os.environ.get("NQ", 48)), 10
rng = np.random.default_rng(77)
comp = (rng.standard_normal((nlists * 13, dim)) * 4).astype(np.float32)
rows, lens = [], []
for L in range(nlists):
    mine = np.arange(L * 13, L * 13 + 13)
    n = 1700
    rows.append((comp[mine[rng.integers(0, 13, n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32))
    lens.append(n)
rows = np.concatenate(rows)
cents = np.stack([rows[sum(lens[:L]):sum(lens[:L + 1])].mean(0) for L in range(nlists)]).astype(np.float32)
q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
lib.check(lib.lib().ndbhip_set_scan_mode(5))
for strategy in (1, 2, 3):
    ix = IvfIndex(dim, nlists); ix.set_centroids(cents)
    ix.load(np.asarray(lens, np.int64), rows, ndbo.tids_from_rows(np.arange(len(rows))))
    lib.check(lib.lib().ndbhip_stats_reset())
    ix.search(q, strategy, nprobe, k, 0)
    print(strategy, lib.stats())
    ix.close()
if os.environ.get("REPEAT"):
    # is the regrouping the same every time? (rows_swept / plane_bytes of the same table, index built again each time)
    seen = {}
    for it in range(int(os.environ["REPEAT"])):
        ix = IvfIndex(dim, nlists); ix.set_centroids(cents)
        ix.load(np.asarray(lens, np.int64), rows, ndbo.tids_from_rows(np.arange(len(rows))))
        lib.check(lib.lib().ndbhip_stats_reset())
        ix.search(q, 1, nprobe, k, 0)
        st = lib.stats()
        seen[(st["rows_swept"], st["plane_bytes"], st["rows_emitted"])] = seen.get((st["rows_swept"], st["plane_bytes"], st["rows_emitted"]), 0) + 1
        ix.close()
        if it % 3 == 0:
            # something else in between, as in the test suite: another table, other sizes
            r2 = rng.standard_normal((3000 + 500 * it, 64)).astype(np.float32)
            ix2 = IvfIndex(64, 4); ix2.set_centroids(r2[:4].copy())
            ix2.load(np.asarray([len(r2), 0, 0, 0], np.int64), r2, ndbo.tids_from_rows(np.arange(len(r2))))
            ix2.search(r2[:40].copy(), 1, 4, 10, 0)
            ix2.close()
    print("repeat:", seen)
