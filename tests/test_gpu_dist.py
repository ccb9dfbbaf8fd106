"""The N > 1 flow end to end on ONE device: two processes share cuda:0, hold one shard each
(IvfIndex.shard / shard_slices), and run neurondb_amd.dist.sharded_search — query-split cluster selection, probe
all-gather, per-shard scan, record all-gather, replay merge — over a gloo group (RCCL refuses two ranks
on one device; the 8-GPU run of bench.py uses the same code over RCCL).  Result == the oracle's."""
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
