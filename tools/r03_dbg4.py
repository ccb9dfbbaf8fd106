import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from bench import make_data, pack_tids
from neurondb_amd import IvfIndex, _lib
from neurondb_amd._lib import check, lib
from neurondb_amd.dist import ShardedSearchBuffers, partition_slices
dev = torch.device("cuda", 0)
_lib.ensure_init(0); _lib.use_torch_stream()
n, dim, nlists, nprobe, k, nq = int(os.environ.get('NVEC', 1000000)), 768, int(os.environ.get('LISTS', 1024)), 32, 10, 4096
W = int(os.environ.get('WORLD', 2))
base = make_data(n, dim, "clustered", nlists, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
q = make_data(nq, dim, "clustered", nlists, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
full = IvfIndex(dim, nlists)
full.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
_, ll, _, _ = full.export(rows=False)
pc = torch.zeros((nq, nprobe), dtype=torch.int32, device=dev)
full.select_clusters_device(q, pc, nprobe); check(lib().ndbhip_synchronize())
pcn = pc.cpu().numpy(); cnt = np.bincount(pcn[pcn >= 0].ravel(), minlength=nlists)
slo, sln, stl = partition_slices(ll, W, cnt)
del base
ix = full.shard_slices(slo[0], sln[0], stl[0])
buf = ShardedSearchBuffers(nq, k, W, dev, nprobe=nprobe)
check(lib().ndbhip_set_option(b"debug_s16", 1))
check(lib().ndbhip_stats_reset())
ix.search_partial_probes_device(q, pc, buf.cand, buf.ncand, buf.total, 1, nprobe, k, 0)
check(lib().ndbhip_synchronize())
print(_lib.stats())
import ctypes as C
hip = C.CDLL("libamdhip64.so")
HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)
thr = torch.full((W, 2 * nq), float("inf"), dtype=torch.float32, device=dev)
state = {"w": 0, "calls": 0}
def record(ptr, cnt_):
    state["calls"] += 1
    lib().ndbhip_synchronize()
    r = hip.hipMemcpy(C.c_void_p(thr[state["w"]].data_ptr()), C.c_void_p(ptr), C.c_size_t(cnt_ * 4), 3)
    return 0
rec_cb = HOOK(record)
check(lib().ndbhip_set_option(b"debug_s16", 0))
shards = [ix] + [None] * (W - 1)
for w in range(W):
    state["w"] = w
    if w: shards[w] = full.shard_slices(slo[w], sln[w], stl[w])
    lib().ndbhip_internal_set_thr_hook(rec_cb)
    shards[w].search_partial_probes_device(q, pc, buf.cand, buf.ncand, buf.total, 1, nprobe, k, 0)
    check(lib().ndbhip_synchronize())
    lib().ndbhip_internal_set_thr_hook(None)
    if w: shards[w].close()
print("hook calls", state["calls"])
tm = thr[:, 0::2].min(dim=0).values.cpu().numpy()
print("min", np.percentile(tm, [0, 50, 90, 99, 100]), "inf count", np.isinf(tm).sum())
thr_min = thr.min(dim=0).values.contiguous()
def give(ptr, cnt_):
    lib().ndbhip_synchronize()
    hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(thr_min.data_ptr()), C.c_size_t(cnt_ * 4), 3)
    return 0
give_cb = HOOK(give)
check(lib().ndbhip_set_option(b"debug_s16", 1))
lib().ndbhip_internal_set_thr_hook(give_cb)
check(lib().ndbhip_stats_reset())
ix.search_partial_probes_device(q, pc, buf.cand, buf.ncand, buf.total, 1, nprobe, k, 0)
check(lib().ndbhip_synchronize())
lib().ndbhip_internal_set_thr_hook(None)
print(_lib.stats())
