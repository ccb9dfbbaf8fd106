#!/usr/bin/env python3
"""What one rank of an N-GPU sharded search does per step, timed on ONE GPU (no collectives): cluster selection
for its query slice, the scan + partial top-k over its 1/N of the lists for all queries, and the replay merge
of N record sets.  Gives the compute side of the strong-scaling curve that bench.py --gpus N will trace."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_data, pack_tids


def timed(fn, sync, reps=5):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    from neurondb_amd.dist import ShardedSearchBuffers, partition_lists, partition_slices, query_slice
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    n, dim, nlists, nprobe, k, nq = int(os.environ.get("NVEC", 1_000_000)), 768, int(os.environ.get("LISTS", 1024)), 32, 10, \
        int(os.environ.get("NQ", 4096))
    comp = int(os.environ.get("COMPONENTS", nlists))
    print(f"table {n} x {dim}, lists {nlists}, {comp} components, {nq} queries per step")
    base = make_data(n, dim, "clustered", comp, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq, dim, "clustered", comp, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    full = IvfIndex(dim, nlists)
    full.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    _, ll, _, _ = full.export(rows=False)
    sync = lambda: check(lib().ndbhip_synchronize())
    ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    t1 = timed(lambda: full.search_device(q, ot, od, oc, 1, nprobe, k, 0), sync)
    print(f"N=1: full step {t1:.3f} ms ({nq / t1 * 1e3:.0f} q/s)")
    qcal = make_data(nq, dim, "clustered", comp, 0.1, 0x5EED0007, 0x5EEDC0DE, dev)     # calibration batch
    del base
    pc = torch.zeros((nq, nprobe), dtype=torch.int32, device=dev)
    full.select_clusters_device(qcal, pc, nprobe)
    sync()
    pcn = pc.cpu().numpy()
    cnt = np.bincount(pcn[pcn >= 0].ravel(), minlength=nlists)
    variants = ((2, "slices"), (4, "slices"), (8, "rows"), (8, "work"), (8, "slices"))
    if os.environ.get("VARIANTS"):                       # e.g. VARIANTS=8:work,8:slices
        variants = tuple((int(v.split(":")[0]), v.split(":")[1]) for v in os.environ["VARIANTS"].split(","))
    for world, how in variants:
        if how == "slices":
            slo, sln, stl = partition_slices(ll, world, cnt)
            loads = sln.sum(1).astype(np.float64)
            work = (sln * (cnt[None] + 1.0)).sum(1)
            worst = int(work.argmax())
            nsplit = int(((sln > 0).sum(0) > 1).sum())
            print(f"  partition by work with {nsplit} lists cut into slices: heaviest rank has "
                  f"{work[worst] / work.sum():.3f} of the work")
            ix = full.shard_slices(slo[worst], sln[worst], stl[worst])
        else:
            owner = partition_lists(ll, world, cnt if how == "work" else None)
            loads = np.bincount(owner, weights=ll, minlength=world)
            work = np.bincount(owner, weights=ll * cnt, minlength=world)
            worst = int(work.argmax())
            print(f"  partition of whole lists by {how}: heaviest rank has {work[worst] / work.sum():.3f} of the work")
            ix = full.shard((owner == worst).astype(np.uint8))
        buf = ShardedSearchBuffers(nq, k, world, dev, nprobe=nprobe)
        lo, hi, s = query_slice(nq, world, 0)
        probes = torch.zeros((nq, nprobe), dtype=torch.int32, device=dev)
        full.select_clusters_device(q, probes, nprobe)
        sync()
        t_sel = timed(lambda: ix.select_clusters_device(q[lo:hi], buf.probes_mine[:hi - lo], nprobe), sync)
        # The ranks of a real run exchange the queries' first thresholds between seeds and sweep (minimum over the ranks:
        # ndbhip_comm_allreduce_min_f32).  Emulated here: every rank's shard is searched once with a hook that records
        # its thresholds, then the heaviest shard is timed with a hook that hands it the minimum over all of them.
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)
        thr = torch.full((world, 2 * nq), float("inf"), dtype=torch.float32, device=dev)
        state = {"w": 0}

        def record(ptr, cnt_):
            lib().ndbhip_synchronize()      # (the library's stream is its own: the seeds must have run)
            hip.hipMemcpy(C.c_void_p(thr[state["w"]].data_ptr()), C.c_void_p(ptr), C.c_size_t(cnt_ * 4), 3)
            return 0
        rec_cb = HOOK(record)
        shards = []
        for w in range(world):
            if how == "slices":
                sh = ix if w == worst else full.shard_slices(slo[w], sln[w], stl[w])
            else:
                sh = ix if w == worst else full.shard((owner == w).astype(np.uint8))
            state["w"] = w
            lib().ndbhip_internal_set_thr_hook(rec_cb)
            sh.search_partial_probes_device(q, probes, buf.cand, buf.ncand, buf.total, 1, nprobe, k, 0)
            sync()
            lib().ndbhip_internal_set_thr_hook(None)
            if w != worst:
                sh.close()
        thr_min = thr.min(dim=0).values.contiguous()

        def give(ptr, cnt_):
            lib().ndbhip_synchronize()
            hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(thr_min.data_ptr()), C.c_size_t(cnt_ * 4), 3)
            return 0
        give_cb = HOOK(give)
        lib().ndbhip_internal_set_thr_hook(give_cb)
        t_scan = timed(lambda: ix.search_partial_probes_device(q, probes, buf.cand, buf.ncand, buf.total, 1, nprobe, k, 0),
                       sync)
        lib().ndbhip_internal_set_thr_hook(None)
        for w in range(world):
            buf.cand_all[w].copy_(buf.cand)
            buf.ncand_all[w].copy_(buf.ncand)
        t_merge = timed(lambda: check(lib().ndbhip_merge_topk_device(
            buf.cand_all.data_ptr(), buf.ncand_all.data_ptr(), buf.total.data_ptr(), world, nq, k, buf.cap,
            buf.out_tids.data_ptr(), buf.out_dist.data_ptr(), buf.out_count.data_ptr())), sync)
        tot = t_sel + t_scan + t_merge
        print(f"N={world}: heaviest shard holds {loads[worst] / loads.sum():.3f} of the rows; select(slice) {t_sel:.3f} ms, "
              f"scan+partial top-k {t_scan:.3f} ms, merge {t_merge:.3f} ms -> {tot:.3f} ms compute per step "
              f"(speed-up {t1 / tot:.2f}, before the two all-gathers)")
        ix.close()


if __name__ == "__main__":
    main()
