#!/usr/bin/env python3
"""The `intended` HNSW at size: build rate, search rate, recall@10 against a float64 brute force.
usage: tools/h2_bench.py N [DIM] [gauss|clustered] [EF ...]   (unit-norm rows, m = 16, ef_construction = 200;
H2_STRATEGY = the operator class's strategy, default 2 = cosine like bench.py's C3 leg)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    from neurondb_amd import HnswIndex, _lib
    from neurondb_amd._lib import check, lib
    n = int(sys.argv[1])
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    kind = sys.argv[3] if len(sys.argv) > 3 else "gauss"
    efs = [int(x) for x in sys.argv[4:]] or [64]
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    import ctypes as C
    nq = int(os.environ.get("H2_NQ", "8192"))
    strategy = int(os.environ.get("H2_STRATEGY", "2"))
    x = torch.empty((n, dim), dtype=torch.float32, device=dev)
    q = torch.empty((nq, dim), dtype=torch.float32, device=dev)
    k1 = 1 if kind == "clustered" else 0
    check(lib().ndbhip_gen_rows_device(k1, 0x5EED0003, 0x5EEDC0DE, 0, n, dim, 1024, 0.1, C.c_void_p(x.data_ptr())))
    check(lib().ndbhip_gen_rows_device(k1, 0x5EED0004, 0x5EEDC0DE, 0, nq, dim, 1024, 0.1, C.c_void_p(q.data_ptr())))
    check(lib().ndbhip_synchronize())
    x = x / x.norm(dim=1, keepdim=True)
    q = q / q.norm(dim=1, keepdim=True)
    r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)
    tids = torch.arange(n, device=dev, dtype=torch.int64)
    ix = HnswIndex(dim, 16)
    check(lib().ndbhip_hnsw_set_intended_select(int(os.environ.get("H2_SELECT", "1"))))
    if os.environ.get("H2_WAVES"):
        check(lib().ndbhip_set_option(b"hnsw_intended_waves", int(os.environ["H2_WAVES"])))
    if os.environ.get("H2_OCC4"):
        check(lib().ndbhip_set_option(b"hnsw_intended_occ4", int(os.environ["H2_OCC4"])))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.build_intended(x, tids, levels, 200, batch_div=int(os.environ.get("H2_BDIV", "16")), batch_max=int(os.environ.get("H2_BMAX", "8192")))
    tb = time.perf_counter() - t0
    st = (C.c_int64 * 6)()
    check(lib().ndbhip_hnsw_build_stats(ix._h, st))
    print(f"build {n} x {dim} {kind}: {tb:.2f} s = {n / tb:.0f} vectors/s, {st[4]} batches (largest {st[5]}), {st[0]} back-links")
    ph = (C.c_ulonglong * 8)()
    check(lib().ndbhip_debug_h2_phases(ph))
    if ph[4]:
        print("  build, phases of block 0's wave, us per expansion: " + ", ".join(
            f"{name} {ph[i] / 100 / ph[4]:.2f}" for i, name in enumerate(("pick", "neighbours + marks", "rows", "offers"))) +
            f"; {ph[4]} expansions, {sum(ph[:4]) / 100 / 1e6:.2f} s of the build in layer searches")
        print(f"  build, block 0's wave: layer searches + clearing {ph[5] / 1e8:.2f} s, sorts {ph[6] / 1e8:.2f} s, selections {ph[7] / 1e8:.2f} s")
    nr = int(os.environ.get("H2_NR", "1000"))
    sims = q[:nr].double() @ x.double().T
    gt = torch.topk(sims, 10, dim=1).indices.cpu().numpy() + 1
    for ef in efs:
        ix.search_intended(q[:256], ef, 10, strategy=strategy)
        ph = (C.c_ulonglong * 8)()
        check(lib().ndbhip_debug_h2_phases(ph))               # (reset: what follows is the timed search alone)
        t0 = time.perf_counter()
        ob, od, oc, oe = ix.search_intended(q, ef, 10, strategy=strategy)
        ts = time.perf_counter() - t0
        rec = float(np.mean([len(set(ob[i, :oc[i]].tolist()) & set(gt[i].tolist())) / 10 for i in range(nr)]))
        print(f"search ef={ef} strategy={strategy}: {nq / ts:.0f} queries/s ({ts * 1e3:.1f} ms per {nq}), recall@10 {rec:.3f}, "
              f"{oe.mean():.0f} evaluations/query")
        if dim % 4 == 0 and dim <= 1024:
            # the same with the walk on fp16 walk rows (ndbhip_hnsw_search_intended_w16_device)
            ix.search_intended(q[:256], ef, 10, walk16=True, strategy=strategy)
            t0 = time.perf_counter()
            wb, wd, wc, we = ix.search_intended(q, ef, 10, walk16=True, strategy=strategy)
            tw = time.perf_counter() - t0
            rec = float(np.mean([len(set(wb[i, :wc[i]].tolist()) & set(gt[i].tolist())) / 10 for i in range(nr)]))
            same = float(np.mean([np.array_equal(wb[i], ob[i]) for i in range(nq)]))
            print(f"   walk on fp16 rows: {nq / tw:.0f} queries/s ({tw * 1e3:.1f} ms per {nq}), recall@10 {rec:.3f}, "
                  f"{we.mean():.0f} evaluations/query (the re-score's {ef} included), {same:.3f} of the queries return the float4 walk's blocks")
        check(lib().ndbhip_debug_h2_phases(ph))
        if ph[4]:
            # a profiling build (make EXTRA=-DNDB_PHASES): block 0's wave, microseconds per expansion
            print("  phases of block 0's wave, us per expansion: " + ", ".join(
                f"{name} {ph[i] / 100 / ph[4]:.2f}" for i, name in enumerate(("pick", "neighbours + marks", "rows", "offers"))) +
                f"; {ph[4]} expansions")


if __name__ == "__main__":
    main()
