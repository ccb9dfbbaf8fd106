#!/usr/bin/env python3
"""HNSW measurement (BASELINE configs[2] shape: m=16, ef_search=64, k=10, cosine on unit-norm rows):
device build rate (ndbhip_hnsw_build_device), batch search rate, distance evaluations and algorithmic
bytes per query, recall@10 vs exact brute force, and parity of a sample against the CPU oracle run on the
device-built graph.   python tools/hnsw_bench.py --nvec 20000 [--dim 768 --m 16 --efc 200]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvec", type=int, default=20000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--ef", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--oracle-sample", type=int, default=50)
    ap.add_argument("--sequential", action="store_true", help="one-wave sequential build (k_hnsw_build)")
    ap.add_argument("--wave-commit", action="store_true", help="optimistic batches with the one-wave commit")
    ap.add_argument("--batch-div", type=int, default=64)
    ap.add_argument("--batch-max", type=int, default=1024)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--graph-hash", action="store_true", help="print a digest of the built graph")
    ap.add_argument("--data", choices=["gauss", "clustered"], default="gauss")
    a = ap.parse_args()
    from neurondb_amd import HnswIndex, _lib
    from neurondb_amd._lib import check, lib
    import ctypes as C
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0003)
    base = torch.randn((a.nvec, a.dim), generator=g, device=dev)
    if a.data == "clustered":                                  # same mixture family as bench.py's default
        cent = torch.randn((1024, a.dim), generator=g, device=dev)
        base = cent[torch.randint(0, 1024, (a.nvec,), generator=g, device=dev)] + 0.1 * base
    base = base / base.norm(dim=1, keepdim=True)               # unit norm: L2, cosine and IP orders coincide (Q1)
    q = torch.randn((a.nq, a.dim), generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    rng = np.random.default_rng(11)
    r = rng.uniform(1e-12, 1.0, a.nvec)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)   # hnsw_am.c:1143-1161
    tids = torch.arange(a.nvec, device=dev, dtype=torch.int64)
    tids = ((tids // 64 >> 16) & 0xFFFF) | ((tids // 64 & 0xFFFF) << 16) | ((tids % 64 + 1) << 32)
    ix = HnswIndex(a.dim, a.m)
    HnswIndex.set_build_mode(0 if a.sequential else (2 if a.wave_commit else 1), a.batch_div, a.batch_max)
    t0 = time.perf_counter()
    check(lib().ndbhip_hnsw_build_device(ix._h, C.c_void_p(base.data_ptr()), C.c_void_p(tids.data_ptr()), a.nvec,
                                         levels.ctypes.data, a.efc))
    tb = time.perf_counter() - t0
    print(f"build: {a.nvec} x {a.dim}, m={a.m}, ef_construction={a.efc}: {tb:.2f} s = {a.nvec / tb:.0f} vectors/s "
          + ("(one wave, sequential inserts)" if a.sequential else f"(optimistic batches: {ix.build_stats()})"))
    if a.graph_hash:
        import hashlib
        e = ix.export()
        hh = hashlib.sha1()
        for key in ("levels", "ncount", "nbrs"):
            hh.update(np.ascontiguousarray(e[key]).tobytes())
        print(f"graph digest: {hh.hexdigest()} entry {e['entry_point']}/{e['entry_level']}")
    if a.build_only:
        return
    ob = torch.zeros((a.nq, a.k), dtype=torch.int32, device=dev)
    od = torch.zeros((a.nq, a.k), dtype=torch.float32, device=dev)
    oc = torch.zeros(a.nq, dtype=torch.int32, device=dev)
    ot = torch.zeros((a.nq, a.k), dtype=torch.int64, device=dev)
    osc = torch.zeros(a.nq, dtype=torch.int64, device=dev)

    def run():
        check(lib().ndbhip_hnsw_search_device(ix._h, C.c_void_p(q.data_ptr()), a.nq, 2, a.ef, a.k,
                                              C.c_void_p(ob.data_ptr()), C.c_void_p(od.data_ptr()),
                                              C.c_void_p(oc.data_ptr()), C.c_void_p(ot.data_ptr()),
                                              C.c_void_p(osc.data_ptr())))
        check(lib().ndbhip_synchronize())
    run()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        run()
    ts = (time.perf_counter() - t0) / reps
    E = float(osc.double().mean())
    bytes_q = E * (48 + 4 * a.dim + 4 * 2 * a.m)              # E x (node header + vector + level-0 slots)
    print(f"search: {a.nq} queries, cosine, ef={a.ef}, k={a.k}: {ts * 1e3:.2f} ms = {a.nq / ts:.0f} q/s; "
          f"{E:.1f} distance evaluations/query = {bytes_q / 1e6:.3f} MB/query algorithmic -> "
          f"{a.nq / ts * bytes_q / 1e9:.1f} GB/s ({a.nq / ts * bytes_q / 8e12 * 100:.2f} % of 8 TB/s: latency-bound walk)")
    # recall vs exact brute force (cosine on unit vectors)
    sims = q.double() @ base.double().T
    gt = torch.topk(sims, a.k, dim=1).indices.cpu().numpy() + 1
    got = ob.cpu().numpy()
    rec = np.mean([len(set(got[i][:int(oc[i])]) & set(gt[i])) / a.k for i in range(a.nq)])
    print(f"recall@{a.k} vs exact: {rec:.3f} (the reference's level-0 walk is BFS-until-ef, quirk Q10)")
    if a.oracle_sample > 0:
        from oracle import ndbo
        e = ix.export()
        vecs = np.zeros((a.nvec + 1, a.dim), np.float32)
        vecs[1:] = base.cpu().numpy()
        og = ndbo.HnswGraph.from_arrays(vecs, e["levels"], e["ncount"], e["nbrs"], None, e["entry_point"],
                                        e["entry_level"], a.m, a.efc)
        qh = q[: a.oracle_sample].cpu().numpy()
        gd = od.cpu().numpy()
        gs = osc.cpu().numpy()
        bad = 0
        t0 = time.perf_counter()
        for i in range(a.oracle_sample):
            eb, ed, ns = og.search(qh[i], 2, a.ef, a.k)
            okk = int(oc[i]) == len(eb) and np.array_equal(got[i, :len(eb)], eb) and \
                np.array_equal(gd[i, :len(eb)].view(np.uint32), ed.view(np.uint32)) and gs[i] == ns
            bad += (not okk)
        tc = (time.perf_counter() - t0) / a.oracle_sample
        print(f"oracle on the same graph: {a.oracle_sample} queries, {bad} mismatches "
              f"(blocks, ranks, float4 bits, evaluation counts); CPU {tc * 1e3:.2f} ms/query single thread")


if __name__ == "__main__":
    main()
