"""N > 1 path on CPU: world_size-2 gloo processes run the product's exchange + merge
(neurondb_amd.dist.gather_and_merge -> ndbhip_merge_topk_host) on per-rank candidate
records; the merged result must equal the oracle's single-process search."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ndbo
from tests.test_merge_host import key_of, partial_records
from tests.util import make_ivf_arrays, oracle_image


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_records(a, img, owner, rank, q, nprobe, k, sel=None):
    """What ndbhip_ivf_search_partial_device emits on `rank` (restated with the oracle's distances)."""
    if sel is None:
        sel = img.select_clusters(q, nprobe)
    ll = a["list_len"]
    off = np.zeros(len(ll) + 1, np.int64)
    off[1:] = np.cumsum(ll)
    dist_l, pos_l, tid_l = [], [], []
    pos = 0
    L = ndbo.lib()
    t64 = (a["tids"]["bi_hi"].astype(np.uint64) | (a["tids"]["bi_lo"].astype(np.uint64) << np.uint64(16)) |
           (a["tids"]["posid"].astype(np.uint64) << np.uint64(32)))
    for c in sel:
        if c < 0:
            continue
        for r in range(off[c], off[c + 1]):
            if isinstance(owner, tuple):          # (lo, length) of partition_slices: this rank's slice of list c
                mine = owner[0][rank, c] <= r - off[c] < owner[0][rank, c] + owner[1][rank, c]
            else:
                mine = owner[c] == rank
            if mine:
                dist_l.append(L.ndbo_ivf_distance(q, a["rows"][r], a["rows"].shape[1], 1))
                pos_l.append(pos)
                tid_l.append(t64[r])
            pos += 1
    rec = partial_records(np.array(dist_l, np.float32), np.array(pos_l, np.uint32), np.array(tid_l, np.uint64), k)
    return rec, pos


def _worker(rank, world, port, seed, slices, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurondb_amd.dist import (ShardedSearchBuffers, gather_and_merge, gather_probes, partition_lists,
                                   partition_slices, query_slice)
    a = make_ivf_arrays(1500, 16, 12, seed=seed, dup_frac=0.2, integer=True)
    img = oracle_image(a)
    if slices:                                   # every list with >= 2 x 8 rows is cut in two
        owner = partition_slices(a["list_len"], world, None, split_frac=0.0, align=8)[:2]
    else:
        owner = partition_lists(a["list_len"], world)
    rng = np.random.default_rng(seed + 1)
    queries = rng.integers(-3, 4, size=(7, 16)).astype(np.float32)      # 7: the last rank's slice is short
    k, nprobe = 10, 5
    buf = ShardedSearchBuffers(len(queries), k, world, "cpu", nprobe=nprobe)
    # cluster selection split by queries: this rank selects for its slice only, the slices are all-gathered
    lo, hi, _ = query_slice(len(queries), world, rank)
    for i in range(lo, hi):
        buf.probes_mine[i - lo] = torch.from_numpy(img.select_clusters(queries[i], nprobe).astype(np.int32))
    probes = gather_probes(buf).numpy()
    ok = all(np.array_equal(probes[i], img.select_clusters(q, nprobe)) for i, q in enumerate(queries))
    for i, q in enumerate(queries):
        rec, total = _rank_records(a, img, owner, rank, q, nprobe, k, sel=probes[i])
        raw = np.zeros((buf.cap,), dtype=rec.dtype)
        raw[:len(rec)] = rec
        buf.cand[i] = torch.from_numpy(raw.view(np.int64).reshape(buf.cap, 2))
        buf.ncand[i] = len(rec)
        buf.total[i] = total
    ot, od, oc = gather_and_merge(buf)
    for i, q in enumerate(queries):
        et, ed, _ = img.search(q, 1, nprobe, k, 0)
        got = ndbo.tids_from_device_u64(ot[i, :len(et)].numpy())
        ok &= int(oc[i]) == len(et)
        ok &= bool(np.array_equal(got, ndbo.tids_to_u64(et)))
        ok &= bool(np.array_equal(od[i, :len(et)].numpy().view(np.uint32), ed.view(np.uint32)))
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slices", [(2, False), (2, True)])
def test_gloo_world2_exchange_and_merge_equals_oracle(world, slices):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 77, slices, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_partition_is_balanced_and_deterministic():
    from neurondb_amd.dist import partition_lists
    rng = np.random.default_rng(0)
    ll = rng.integers(0, 5000, 1024)
    o1, o2 = partition_lists(ll, 8), partition_lists(ll, 8)
    assert np.array_equal(o1, o2)
    loads = np.bincount(o1, weights=ll, minlength=8)
    assert loads.max() - loads.min() <= ll.max()


def test_partition_slices_covers_every_list_once_and_balances_work():
    from neurondb_amd.dist import partition_slices
    rng = np.random.default_rng(1)
    ll = rng.integers(0, 3000, 1024)
    ll[5], ll[7] = 28000, 0                               # one long, popular list; one empty list
    pc = rng.integers(0, 400, 1024)
    pc[5] = 3000
    for world in (1, 2, 8):
        lo, ln, tail = partition_slices(ll, world, pc)
        lo2, ln2, tail2 = partition_slices(ll, world, pc)
        assert np.array_equal(lo, lo2) and np.array_equal(ln, ln2) and np.array_equal(tail, tail2)
        assert (ln.sum(0) == ll).all() and (tail.sum(0) == 1).all()
        for l in np.nonzero((ln > 0).sum(0) > 1)[0]:      # slices of a cut list tile it, on 64-row boundaries
            parts = sorted((int(lo[r, l]), int(ln[r, l])) for r in range(world) if ln[r, l] > 0)
            pos = 0
            for a, n in parts:
                assert a == pos and a % 64 == 0
                pos += n
            assert pos == ll[l]
            assert tail[np.argmax(lo[:, l] + ln[:, l]), l] == 1
        work = (ln * (pc[None] + 1.0)).sum(1)
        assert work.max() / work.sum() < 1.0 / world + 0.01
        if world == 8:
            assert (ln[:, 5] > 0).all()                   # the 28 k-row list is shared by all ranks


def _comm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurondb_amd.dist import init_library_comm
    try:
        init_library_comm()
        outcome = "ok"
    except RuntimeError as e:
        outcome = "raised: " + str(e)[:60]
    # the ranks are still in step: one more collective pairs up and carries the right values
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    ret[rank] = (outcome, int(t))
    dist.destroy_process_group()


def test_library_communicator_setup_fails_on_every_rank_or_on_none():
    """init_library_comm without a device: rank 0 cannot draw an RCCL unique id.  It must still take part in the
    broadcast and the agreement round, so that every rank raises (bench.py then falls back to the torch exchange on
    all of them) instead of rank 1 waiting in a broadcast rank 0 never joins."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_comm_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    outcomes = {ret[r][0].split(":")[0] for r in range(world)}
    assert len(outcomes) == 1                       # the same branch everywhere
    assert all(ret[r][1] == 3 for r in range(world))
