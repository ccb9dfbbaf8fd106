"""The N > 1 flow end to end on ONE device: two processes share cuda:0, hold one shard each
(IvfIndex.shard / shard_slices), and run the sharded search — query-split cluster selection, probe all-gather,
per-shard scan, record all-gather, replay merge.  Result == the oracle's.
  * through the C ABI (ndbhip_ivf_search_sharded, csrc/ndbhip_comm.cpp) over the library's shared-memory
    transport (RCCL refuses two ranks on one device), and in one process over a world-1 RCCL communicator — the
    same function bench.py --gpus N runs over RCCL;
  * through neurondb_amd.dist.sharded_search over a gloo group (the torch.distributed form of the same steps)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ndbo
from tests.util import make_ivf_arrays, oracle_image, oracle_search_batch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, slices, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neurondb_amd import IvfIndex, _lib
        from neurondb_amd.dist import ShardedSearchBuffers, partition_lists, partition_slices, sharded_search
        _lib.ensure_init(0)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        # kernels and collectives are ordered by sharing torch's stream (what bench.py does)
        _lib.use_torch_stream()
        a = make_ivf_arrays(4000, 64, 20, seed=51, dup_frac=0.1)
        img = oracle_image(a)
        rng = np.random.default_rng(52)
        nq = 150 if slices else 29                       # 150: the screened scan (>= 128 queries) in partial mode
        q = a["rows"][rng.integers(0, len(a["rows"]), nq)] + rng.standard_normal((nq, 64)).astype(np.float32) * 0.05
        q = np.ascontiguousarray(q, dtype=np.float32)
        k, nprobe = 10, 5
        full = IvfIndex(64, 20)
        full.set_centroids(a["centroids"])
        full.load(a["list_len"], a["rows"], a["tids"])
        if slices:
            lo, ln, tail = partition_slices(a["list_len"], world, None, split_frac=0.0, align=16)
            ix = full.shard_slices(lo[rank], ln[rank], tail[rank])
        else:
            owner = partition_lists(a["list_len"], world)
            ix = full.shard((owner == rank).astype(np.uint8))
        full.close()
        buf = ShardedSearchBuffers(len(q), k, world, dev, nprobe=nprobe)
        dq = torch.from_numpy(q).to(dev)
        ot, od, oc = sharded_search(ix, dq, buf, 1, nprobe, k, 0, rank=rank)
        _lib.check(_lib.lib().ndbhip_synchronize())
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
        ok = bool(np.array_equal(oc.cpu().numpy(), ec))
        ok &= bool(np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et))
        ok &= bool(np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32)))
        ret[rank] = ok
    except Exception as e:                                      # surfaced by the parent's assert
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("slices", [False, True])
def test_two_ranks_one_device_sharded_search_equals_oracle(slices):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), slices, ret), nprocs=world, join=True)
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def _worker_c(rank, world, name, slices, ret, rccl_id=None):
    """rccl_id None: both ranks on cuda:0 over the shared-memory transport; else a Manager dict through which rank 0 hands
    the other ranks RCCL's unique id: rank r on cuda:r, the communicator bench.py --gpus N uses."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        from neurondb_amd import IvfIndex, _lib
        from neurondb_amd.dist import partition_lists, partition_slices
        di = rank if rccl_id is not None else 0
        _lib.ensure_init(di)
        dev = torch.device("cuda", di)
        torch.cuda.set_device(dev)
        _lib.use_torch_stream()
        a = make_ivf_arrays(6000, 96, 20, seed=61, dup_frac=0.1)
        a["rows"][100:420] = a["rows"][100]              # 320 equal rows: more ties than a query's survivor buffer holds
        img = oracle_image(a)
        rng = np.random.default_rng(62)
        nq, k, nprobe = 157, 10, 5                      # ragged query slices; >= 128 queries: the fp16 matrix-core screen
        q = a["rows"][rng.integers(0, len(a["rows"]), nq)] + rng.standard_normal((nq, 96)).astype(np.float32) * 0.05
        q = np.ascontiguousarray(q, dtype=np.float32)
        q[5] = q[6] = a["rows"][100]                    # ... for these queries: they go to the exact path alone, on each rank
        q[7] = 0.0                                      # a zero query (cosine distance exactly 1 to everything)
        full = IvfIndex(96, 20)
        full.set_centroids(a["centroids"])
        full.load(a["list_len"], a["rows"], a["tids"])
        if slices:
            lo, ln, tail = partition_slices(a["list_len"], world, None, split_frac=0.0, align=16)
            ix = full.shard_slices(lo[rank], ln[rank], tail[rank])
        else:
            owner = partition_lists(a["list_len"], world)
            ix = full.shard((owner == rank).astype(np.uint8))
        full.close()
        if rccl_id is None:
            _lib.check(_lib.lib().ndbhip_comm_init_shm(name.encode(), rank, world, 1 << 20))
        else:
            import ctypes as C
            import time as _t
            ident = (C.c_ubyte * 128)()
            if rank == 0:
                _lib.check(_lib.lib().ndbhip_comm_unique_id(C.byref(ident)))
                rccl_id["id"] = bytes(ident)
            else:
                t0 = _t.time()
                while "id" not in rccl_id:
                    if _t.time() - t0 > 120:
                        raise TimeoutError("rank 0 never published the RCCL id")
                    _t.sleep(0.01)
                C.memmove(ident, rccl_id["id"], 128)
            _lib.check(_lib.lib().ndbhip_comm_init(C.byref(ident), rank, world))
        assert _lib.lib().ndbhip_comm_world() == world and _lib.lib().ndbhip_comm_rank() == rank
        dq = torch.from_numpy(q).to(dev)
        ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        ok = True
        for strategy in (1, 3, 2):
            ix.search_sharded_device(dq, ot, od, oc, strategy, nprobe, k, 0)
            _lib.check(_lib.lib().ndbhip_synchronize())
            et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k)
            ok &= bool(np.array_equal(oc.cpu().numpy(), ec))
            ok &= bool(np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et))
            ok &= bool(np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32)))
        _lib.check(_lib.lib().ndbhip_comm_destroy())
        ret[rank] = ok
    except Exception as e:
        ret[rank] = f"{type(e).__name__}: {e}"


@pytest.mark.parametrize("slices", [False, True])
def test_c_abi_sharded_search_two_ranks_over_shared_memory(slices):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_c, args=(world, f"/ndbhip_test_{os.getpid()}_{int(slices)}", slices, ret), nprocs=world, join=True)
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


@pytest.mark.parametrize("slices", [False, True])
def test_c_abi_sharded_search_two_ranks_over_rccl_on_two_devices(slices):
    """The same two-rank search with one rank per DEVICE over RCCL (ncclCommInitRank with two ranks, the all-gathers and the
    all-reduce(min) of ndbhip_comm.cpp between two GPUs): what bench.py --gpus N times.  Needs two devices: skipped on the
    one-GPU boxes the round's tests run on, there for the day a multi-GPU box runs them."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} device(s): two ranks over RCCL need two")
    world = 2
    mgr = mp.Manager()
    ret, rid = mgr.dict(), mgr.dict()
    mp.spawn(_worker_c, args=(world, "", slices, ret, rid), nprocs=world, join=True)
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_c_abi_sharded_search_over_a_world_1_rccl_communicator():
    """ncclCommInitRank / ncclAllGather really run (one rank): the RCCL transport of ndbhip_comm.cpp, as far as
    one device can exercise it."""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    _lib.ensure_init(0)
    L = _lib.lib()
    a = make_ivf_arrays(5000, 64, 16, seed=71, dup_frac=0.05)
    img = oracle_image(a)
    rng = np.random.default_rng(72)
    nq, k, nprobe = 140, 10, 6
    q = np.ascontiguousarray(a["rows"][rng.integers(0, 5000, nq)] +
                             0.05 * rng.standard_normal((nq, 64)), dtype=np.float32)
    ix = IvfIndex(64, 16)
    ix.set_centroids(a["centroids"])
    ix.load(a["list_len"], a["rows"], a["tids"])
    ident = (C.c_ubyte * 128)()
    _lib.check(L.ndbhip_comm_unique_id(C.byref(ident)))
    _lib.check(L.ndbhip_comm_init(C.byref(ident), 0, 1))
    try:
        dev = torch.device("cuda", 0)
        dq = torch.from_numpy(q).to(dev)
        ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        # the raw collective first
        src = torch.arange(1000, dtype=torch.int32, device=dev)
        dst = torch.zeros(1000, dtype=torch.int32, device=dev)
        _lib.check(L.ndbhip_comm_allgather(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 4000))
        ix.search_sharded_device(dq, ot, od, oc, 1, nprobe, k, 0)
        _lib.check(L.ndbhip_synchronize())
        assert torch.equal(src, dst)
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
        assert np.array_equal(oc.cpu().numpy(), ec)
        assert np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et)
        assert np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    finally:
        _lib.check(L.ndbhip_comm_destroy())
        ix.close()


def _worker_build(rank, world, name, n, dim, nlists, ret):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        from neurondb_amd import IvfIndex, _lib
        from bench import pack_tids
        _lib.ensure_init(0)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        _lib.use_torch_stream()
        rng = np.random.default_rng(81)
        cents = rng.standard_normal((nlists, dim)).astype(np.float32) * 3
        base = (cents[rng.integers(0, nlists, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
        base[n // 3] = base[n // 3 + 1]                          # a duplicate row across the cut is still two rows
        # oracle: the single-process build of the whole table
        img, asg, iters = ndbo.build_ivf_image(base, nlists, max_iter=50)
        cut = [0, 5000 + (n - 5000) // 3, n][: world + 1] if world == 2 else [0, n]   # uneven slices; rank 0 holds the sample
        lo, hi = cut[rank], cut[rank + 1]
        _lib.check(_lib.lib().ndbhip_comm_init_shm(name.encode(), rank, world, 64 << 20))
        ix = IvfIndex(dim, nlists)
        d_rows = torch.from_numpy(base[lo:hi]).to(dev)
        d_tids = pack_tids(torch.arange(lo, hi, device=dev))
        if os.environ.get("NDBHIP_TEST_TRACE"):
            _lib.check(_lib.lib().ndbhip_set_option(b"debug_build", 1))
        it, owned = ix.build_sharded_device(d_rows, d_tids, 50)
        _lib.check(_lib.lib().ndbhip_synchronize())
        why = []
        tr = (lambda m: print(f"[rank {rank}] {m}", file=__import__("sys").stderr, flush=True)) \
            if os.environ.get("NDBHIP_TEST_TRACE") else (lambda m: None)
        tr("built")
        try:                                                     # (no exception may keep this rank from the collectives below)
            cent, ll, rows, tids = ix.export()
            glen = np.diff(img.list_off)
            if it != iters:
                why.append(f"iterations {it} != {iters}")
            if not np.array_equal(cent.view(np.uint32), img.centroids.view(np.uint32)):
                why.append("centroids")
            if not np.array_equal(ll, np.where(owned != 0, glen, 0)):
                why.append("list lengths")
            # the rows of the lists held here, list by list in heap order = the oracle image's rows of those lists
            keep = np.concatenate([np.arange(img.list_off[L], img.list_off[L + 1]) for L in range(nlists) if owned[L]] +
                                  [np.zeros(0, np.int64)]).astype(np.int64)
            if not np.array_equal(rows.view(np.uint32), img.vecs[keep].view(np.uint32)):
                why.append("rows")
            if not np.array_equal(ndbo.tids_to_u64(tids), ndbo.tids_to_u64(img.tids[keep])):
                why.append("tids")
        except Exception as e:
            why.append(f"{type(e).__name__}: {e}")
        ok = not why
        tr(f"checked: {why}")
        # and the shard answers like one: sharded search over both ranks = the oracle on the whole image
        nq, k, nprobe = 150, 10, 6
        q = np.ascontiguousarray(base[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim)), dtype=np.float32)
        dq = torch.from_numpy(q).to(dev)
        ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        tr("searching")
        ix.search_sharded_device(dq, ot, od, oc, 1, nprobe, k, 0)
        _lib.check(_lib.lib().ndbhip_synchronize())
        tr("searched")
        et, ed, ec, _ = oracle_search_batch(img, q, 1, nprobe, k)
        ok &= bool(np.array_equal(oc.cpu().numpy(), ec))
        ok &= bool(np.array_equal(ndbo.tids_from_device_u64(ot.cpu().numpy()), et))
        ok &= bool(np.array_equal(od.cpu().numpy().view(np.uint32), ed.view(np.uint32)))
        _lib.check(_lib.lib().ndbhip_comm_destroy())
        ret[rank] = (bool(ok), int(owned.sum()), int(ix.nrows)) if ok else f"mismatch: {why}"
    except Exception as e:
        import sys
        import traceback
        ret[rank] = f"{type(e).__name__}: {e} {traceback.format_exc()[-600:]}"
        print(f"[rank {rank}] {ret[rank]}", file=sys.stderr, flush=True)
        raise                                                    # a dead rank ends mp.spawn instead of leaving its peer in a collective


def test_c_abi_sharded_build_two_ranks_over_shared_memory():
    """ndbhip_ivf_build_sharded: rank 0 holds the first rows (and the k-means sample), rank 1 the rest; the
    centroids are the single-process build's, every list ends up whole on one rank in heap order, and a sharded
    search over the two shards equals the oracle on the whole table."""
    world, n, dim, nlists = 2, 9000, 64, 48
    mgr = mp.Manager()
    ret = mgr.dict()
    try:
        mp.spawn(_worker_build, args=(world, f"/ndbhip_testb_{os.getpid()}", n, dim, nlists, ret), nprocs=world, join=True)
    except Exception as e:
        raise AssertionError(f"{e}; {dict(ret)}")
    assert all(isinstance(ret.get(r), tuple) and ret[r][0] is True for r in range(world)), dict(ret)
    assert ret[0][1] + ret[1][1] == nlists and ret[0][2] + ret[1][2] == n
    assert min(ret[0][2], ret[1][2]) > n // 4                  # the deal balances rows, not lists


def test_c_abi_alltoallv_and_sharded_build_over_a_world_1_rccl_communicator():
    """ncclSend / ncclRecv to self inside a group, and the sharded build's whole path on a one-rank communicator
    (it must give the single-process build)."""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    from bench import pack_tids
    _lib.ensure_init(0)
    L = _lib.lib()
    ident = (C.c_ubyte * 128)()
    _lib.check(L.ndbhip_comm_unique_id(C.byref(ident)))
    _lib.check(L.ndbhip_comm_init(C.byref(ident), 0, 1))
    try:
        dev = torch.device("cuda", 0)
        src = torch.arange(5000, dtype=torch.int32, device=dev)
        dst = torch.zeros(5000, dtype=torch.int32, device=dev)
        off = (C.c_size_t * 2)(0, 20000)
        _lib.check(L.ndbhip_comm_alltoallv(C.c_void_p(src.data_ptr()), off, C.c_void_p(dst.data_ptr()), off))
        _lib.check(L.ndbhip_synchronize())
        assert torch.equal(src, dst)
        rng = np.random.default_rng(5)
        base = rng.standard_normal((6000, 32)).astype(np.float32)
        img, _, iters = ndbo.build_ivf_image(base, 20, max_iter=50)
        ix = IvfIndex(32, 20)
        it, owned = ix.build_sharded_device(torch.from_numpy(base).to(dev), pack_tids(torch.arange(6000, device=dev)), 50)
        cent, ll, rows, tids = ix.export()
        assert it == iters and owned.all()
        assert np.array_equal(ll, np.diff(img.list_off)) and np.array_equal(rows.view(np.uint32), img.vecs.view(np.uint32))
    finally:
        _lib.check(L.ndbhip_comm_destroy())
