#!/usr/bin/env python3
"""Single-query latency through the host-pointer C ABI (ndbhip_ivf_search, nq = 1) — what one
PostgreSQL backend sees per amgettuple: H2D query, 4 kernels, D2H results, stream sync.
Usage: python tools/latency.py [--nvec 1000000 --dim 768 --lists 1024 --probes 32]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_data, pack_tids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvec", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--lists", type=int, default=1024)
    ap.add_argument("--probes", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--n", type=int, default=1000)
    a = ap.parse_args()
    from neurondb_amd import IvfIndex, _lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    base = make_data(a.nvec, a.dim, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(a.n + 20, a.dim, "clustered", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev).cpu().numpy()
    ix = IvfIndex(a.dim, a.lists)
    ix.build_device(base, pack_tids(torch.arange(a.nvec, device=dev)), 50)
    for i in range(20):
        ix.search(q[i:i + 1], 1, a.probes, a.k)
    _lib.check(_lib.lib().ndbhip_set_option(b"slow_call_log", 1000))      # calls over 1 ms say where the time went (stderr)
    import gc
    gc.collect()
    if os.environ.get("LAT_NOGC"):
        gc.disable()
    ts = []
    for i in range(20, 20 + a.n):
        t0 = time.perf_counter()
        ix.search(q[i:i + 1], 1, a.probes, a.k)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    slow = np.argsort(ts)[-5:][::-1]
    print("  slowest calls (index: us): " + ", ".join(f"{int(i)}: {ts[i]:.0f}" for i in slow) +
          f"; calls over 1 ms: {int((ts > 1000).sum())} of {len(ts)}")
    print(f"single-query ndbhip_ivf_search latency over {a.n} queries: p50 {np.percentile(ts, 50):.0f} us, "
          f"p90 {np.percentile(ts, 90):.0f} us, p99 {np.percentile(ts, 99):.0f} us, mean {ts.mean():.0f} us "
          f"({a.nvec}x{a.dim}, lists={a.lists}, probes={a.probes}, k={a.k})")
    for nq in (8, 64, 256):
        ix.search(q[:nq], 1, a.probes, a.k)          # (the first screened batch lays the planes out: not what is timed)
        t0 = time.perf_counter()
        reps = 20
        for r in range(reps):
            ix.search(q[:nq], 1, a.probes, a.k)
        dt = (time.perf_counter() - t0) / reps
        print(f"  batch of {nq:4d} host-pointer queries: {dt * 1e3:.3f} ms  ({nq / dt:.0f} q/s incl. PCIe + sync)")


if __name__ == "__main__":
    main()
