"""Multi-GPU driver for sharded IVF search: one process per GPU, lists partitioned
over the ranks, per-rank candidate records all-gathered (RCCL over xGMI when the
tensors live on the GPU, gloo on CPU tensors in the tests) and merged by replaying
the reference's selection sort on the union (ndbhip_merge_topk_*).

The exchange the path needs is this all-gather of nq x 3k x 16 B per rank;
there is no reduction and no all-to-all.  A second, smaller all-gather (nq x nprobe x 4 B)
lets the ranks split the centroid scan + ivfSelectClusters by queries instead of all
repeating it for every query (`split_select`)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check, lib

CAND_WORDS = 2          # one ndbhip_cand = 16 bytes = 2 x int64


def partition_lists(list_len, world: int, probe_count=None) -> np.ndarray:
    """Balanced list -> rank map (longest-processing-time first). Deterministic: every rank computes the same
    map from replicated inputs.  A list costs len x (queries that probe it) per batch, so with `probe_count`
    (how often each list was probed by a calibration batch, e.g. ndbhip_ivf_select_clusters on recent queries)
    the WORK is balanced; without it the rows are (popular lists then overload their rank: at 1M x 768,
    lists = 1024 the 28 k-row list alone is 10 % of a batch's sums)."""
    list_len = np.asarray(list_len, dtype=np.int64)
    weight = list_len.astype(np.float64)
    if probe_count is not None:
        weight = weight * (np.asarray(probe_count, dtype=np.float64) + 1.0)
    order = np.argsort(-weight, kind="stable")
    load = np.zeros(world, dtype=np.float64)
    owner = np.zeros(len(list_len), dtype=np.int32)
    for l in order:
        r = int(load.argmin())
        owner[l] = r
        load[r] += weight[l]
    return owner


def partition_slices(list_len, world: int, probe_count=None, split_frac: float = 0.25, align: int = 64):
    """Balanced partition that may split a list: returns (lo, length, tail), each [world, nlists] — rank r holds
    positions [lo[r, L], lo[r, L] + length[r, L]) of list L and takes the appends to the lists with tail[r, L]
    (IvfIndex.shard_slices).  A list whose work exceeds split_frac x (total / world) is cut into `world` slices
    on `align`-row boundaries (the scan's tile height), one per rank; the others go whole, longest-processing-
    time first.  Without splitting, the heaviest list bounds the speed-up (1M x 768, lists = 1024: one list is
    22 % of a batch's sums, so 8 ranks gain 4x at most).  Deterministic from replicated inputs."""
    list_len = np.asarray(list_len, dtype=np.int64)
    nl = len(list_len)
    weight = list_len.astype(np.float64)
    if probe_count is not None:
        weight = weight * (np.asarray(probe_count, dtype=np.float64) + 1.0)
    lo = np.zeros((world, nl), dtype=np.int64)
    length = np.zeros((world, nl), dtype=np.int64)
    tail = np.zeros((world, nl), dtype=np.uint8)
    load = np.zeros(world, dtype=np.float64)
    target = weight.sum() / max(world, 1)
    order = np.argsort(-weight, kind="stable")
    turn = 0
    for l in order:
        n = int(list_len[l])
        if world > 1 and weight[l] > split_frac * target and n >= world * align:
            cuts = [(n * j // world) // align * align for j in range(world)] + [n]
            for j in range(world):
                r = (j + turn) % world          # rotate, so the (longer) last slice does not always hit one rank
                lo[r, l], length[r, l] = cuts[j], cuts[j + 1] - cuts[j]
                load[r] += weight[l] * length[r, l] / n
                tail[r, l] = j == world - 1
            turn += 1
        else:
            r = int(load.argmin())
            length[r, l] = n
            tail[r, l] = 1
            load[r] += weight[l]
    return lo, length, tail


def init_library_comm(group=None, device=None):
    """Create the library's own RCCL communicator (ndbhip_comm_init) for the ranks of a torch.distributed group:
    rank 0 draws the unique id, torch.distributed only carries its 128 bytes to the others.  After this
    IvfIndex.search_sharded_device exchanges inside the C library — the path a PostgreSQL backend uses."""
    import ctypes as C
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu")
    # Every rank takes the same number of collectives whether or not a step fails: rank 0 always broadcasts (a
    # status byte in front of the id), and the outcome of ndbhip_comm_init is all-reduced, so that the ranks agree
    # on success or failure and a caller's fallback runs on all of them or on none.
    msg = torch.zeros(129, dtype=torch.uint8)
    err = None
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        try:
            check(lib().ndbhip_comm_unique_id(C.byref(buf)))
            msg[0] = 1
            msg[1:] = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        except Exception as e:                                # noqa: BLE001 — reported below, after the broadcast
            err = e
    msg = msg.to(dev)
    dist.broadcast(msg, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    host = msg.cpu()
    ok = torch.zeros(1, dtype=torch.int32)
    if int(host[0]) == 1:
        try:
            check(lib().ndbhip_comm_init(C.c_char_p(bytes(host[1:].numpy().tobytes())), rank, world))
            ok[0] = 1
        except Exception as e:                                # noqa: BLE001
            err = e
    ok = ok.to(dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.cpu()[0]) != 1:
        try:
            lib().ndbhip_comm_destroy()                       # ranks that did get a communicator give it back
        except Exception:                                     # noqa: BLE001
            pass
        raise RuntimeError(f"the library's communicator could not be created on every rank"
                           + (f" (this rank: {type(err).__name__}: {err})" if err else " (another rank failed)"))
    return rank, world


def partial_cap(k: int) -> int:
    return 3 * k            # NDBHIP_PARTIAL_CAP


def query_slice(nq: int, world: int, rank: int):
    """Queries [lo, hi) whose probes `rank` selects; every slice has the same padded length ceil(nq / world)."""
    s = -(-nq // world)
    return min(rank * s, nq), min((rank + 1) * s, nq), s


class ShardedSearchBuffers:
    """Pre-allocated exchange buffers for batches of nq queries."""

    def __init__(self, nq: int, k: int, world: int, device, nprobe: int = 0):
        if torch.device(device).type == "cuda":
            # the library's kernels, torch's copies and the collectives must run in one order: one stream
            from . import _lib
            _lib.use_torch_stream()
        cap = partial_cap(k)
        self.nq, self.k, self.world, self.cap = nq, k, world, cap
        self.nprobe = nprobe
        if nprobe:
            s = -(-nq // world)
            self.probes_mine = torch.zeros((s, nprobe), dtype=torch.int32, device=device)
            self.probes_all = torch.zeros((world * s, nprobe), dtype=torch.int32, device=device)
        self.cand = torch.zeros((nq, cap, CAND_WORDS), dtype=torch.int64, device=device)
        self.ncand = torch.zeros(nq, dtype=torch.int32, device=device)
        self.total = torch.zeros(nq, dtype=torch.int64, device=device)
        self.cand_all = torch.zeros((world, nq, cap, CAND_WORDS), dtype=torch.int64, device=device)
        self.ncand_all = torch.zeros((world, nq), dtype=torch.int32, device=device)
        self.out_tids = torch.zeros((nq, k), dtype=torch.int64, device=device)
        self.out_dist = torch.zeros((nq, k), dtype=torch.float32, device=device)
        self.out_count = torch.zeros(nq, dtype=torch.int32, device=device)


def gather_and_merge(buf: ShardedSearchBuffers, group=None):
    """all-gather the ranks' records, then merge.  Device tensors: RCCL + the HIP merge kernel
    (asynchronous on the current stream).  CPU tensors: gloo + ndbhip_merge_topk_host."""
    if buf.world > 1 or (dist.is_available() and dist.is_initialized()):
        # output = ranks concatenated along dim 0 (the layout both RCCL and gloo accept)
        dist.all_gather_into_tensor(buf.cand_all.view(buf.world * buf.nq, buf.cap, CAND_WORDS), buf.cand, group=group)
        dist.all_gather_into_tensor(buf.ncand_all.view(buf.world * buf.nq), buf.ncand, group=group)
    else:
        buf.cand_all[0].copy_(buf.cand)
        buf.ncand_all[0].copy_(buf.ncand)
    fn = lib().ndbhip_merge_topk_device if buf.cand.is_cuda else lib().ndbhip_merge_topk_host
    check(fn(buf.cand_all.data_ptr(), buf.ncand_all.data_ptr(), buf.total.data_ptr(), buf.world, buf.nq, buf.k,
             buf.cap, buf.out_tids.data_ptr(), buf.out_dist.data_ptr(), buf.out_count.data_ptr()))
    return buf.out_tids, buf.out_dist, buf.out_count


def gather_probes(buf: ShardedSearchBuffers, group=None):
    """all-gather the per-rank probe slices (buf.probes_mine) -> probes of all nq queries [nq, nprobe]."""
    if buf.world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_gather_into_tensor(buf.probes_all, buf.probes_mine, group=group)
    else:
        buf.probes_all.copy_(buf.probes_mine)
    return buf.probes_all[:buf.nq]


def sharded_search(index, d_queries, buf: ShardedSearchBuffers, strategy=1, nprobe=10, k=10, max_candidates=0,
                   group=None, rank=None):
    """One batch on this rank's shard (`index` = IvfIndex.shard(owned)) + exchange + merge.
    With buffers built for `nprobe` (and world > 1) the ranks split cluster selection by queries."""
    if buf.nprobe == nprobe and buf.world > 1:
        r = dist.get_rank(group) if rank is None else rank
        lo, hi, _ = query_slice(buf.nq, buf.world, r)
        if hi > lo:
            index.select_clusters_device(d_queries[lo:hi], buf.probes_mine[:hi - lo], nprobe)
        probes = gather_probes(buf, group)
        index.search_partial_probes_device(d_queries, probes, buf.cand, buf.ncand, buf.total, strategy, nprobe, k,
                                           max_candidates)
    else:
        index.search_partial_device(d_queries, buf.cand, buf.ncand, buf.total, strategy, nprobe, k, max_candidates)
    return gather_and_merge(buf, group)
