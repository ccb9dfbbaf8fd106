import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from neurondb_amd import IvfIndex, _lib
from oracle import ndbo
_lib.ensure_init(0); _lib.use_torch_stream()
L=_lib.lib(); check=_lib.check
rng = np.random.default_rng(77)
dim, nlists, per = 96, 24, 300
cents = (rng.standard_normal((nlists, dim)) * 5).astype(np.float32)
rows, lens = [], []
for Li in range(nlists):
    n = 1 if Li == 5 else (120 if Li == 7 else per)
    r = cents[Li] + 0.2 * rng.standard_normal((n, dim)).astype(np.float32)
    if Li == 7:
        u = rng.standard_normal((n, dim)).astype(np.float32)
        r = (cents[Li] + 8.0 * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
    rows.append(r.astype(np.float32)); lens.append(n)
rows = np.concatenate(rows)
rows[per * 9 + 3, 2] = np.nan
ix = IvfIndex(dim, nlists, device=0)
ix.set_centroids(cents); ix.load(np.asarray(lens, np.int64), rows, ndbo.tids_from_rows(np.arange(len(rows))))
nq, k, nprobe = 200, 10, 12
src = rng.integers(0, nlists, nq)
q = (cents[src] + 0.2 * rng.standard_normal((nq, dim))).astype(np.float32)
q[0] = cents[3]; q[1] = rows[per * 7 + 2]
check(L.ndbhip_set_scan_mode(5))
check(L.ndbhip_set_option(b"debug_s16", 1))
for prune in (1, 0):
    check(L.ndbhip_set_option(b"screen16_prune", prune))
    check(L.ndbhip_stats_reset())
    ix.search(q, 1, nprobe, k)
    st=_lib.stats(); print("prune", prune, {k2: st[k2] for k2 in ("screen16_batches","screen16_fallbacks","rows_emitted","rows_rescored")}, flush=True)
