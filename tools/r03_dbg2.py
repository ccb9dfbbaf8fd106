import numpy as np, sys
sys.path.insert(0, '.')
from tests.util import make_ivf_arrays, oracle_image, oracle_search_batch
from neurondb_amd import IvfIndex, _lib
from oracle import ndbo
_lib.ensure_init()
L = _lib.lib()
dim, nlists = 64, 300
n = 6 * nlists
a = make_ivf_arrays(n, dim, nlists, seed=nlists, dup_frac=0.05)
rng = np.random.default_rng(nlists + 1)
cent = a["centroids"]
cent[7] = cent[3]; cent[nlists - 1] = cent[nlists // 2]; cent[11, 5] = np.nan; cent[13, 2] = 3.0e38
img = oracle_image(a)
ix = IvfIndex(dim, nlists); ix.set_centroids(a["centroids"]); ix.load(a["list_len"], a["rows"], a["tids"])
nq = 160
q = rng.standard_normal((nq, dim)).astype(np.float32)
q[:40] = cent[rng.integers(0, nlists, 40)]
q[:40] = np.where(np.isfinite(q[:40]) & (np.abs(q[:40]) < 1e30), q[:40], 0.0)
q[-1] = 0.0
_lib.check(L.ndbhip_set_scan_mode(5))
for opt in (1, 0):
    _lib.check(L.ndbhip_set_option(b"cent_screen16", opt))
    t, d, c = ix.search(q, 1, 7, 10, 0)
    et, ed, ec, _ = oracle_search_batch(img, q, 1, 7, 10, 0)
    bad = [i for i in range(nq) if c[i] != ec[i] or not np.array_equal(ndbo.tids_to_u64(t[i, :c[i]]), et[i, :ec[i]])]
    print("cent_screen16", opt, "bad queries", bad)
    for i in bad[:3]:
        print(i, c[i], ec[i], d[i], ed[i])
        dd = np.sqrt(((q[i][None, :].astype(np.float64) - cent.astype(np.float64)) ** 2).sum(1))
        o = np.argsort(dd, kind="stable")[:9]
        print("nearest centroids", o, dd[o], "list lens", a["list_len"][o])
