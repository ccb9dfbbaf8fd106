#!/usr/bin/env python3
"""Per-query scan vs query-grouped scan across batch sizes (device-resident queries), to place the
auto-dispatch threshold.  python tools/batch_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import make_data, pack_tids


def main():
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    n, dim, nlists, nprobe, k = 1_000_000, 768, 1024, 32, 10
    base = make_data(n, dim, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(16384, dim, "clustered", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, nlists)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    for nq in (1, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 16384):
        ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        oc = torch.zeros(nq, dtype=torch.int32, device=dev)
        line = f"nq={nq:5d}"
        for mode, name in ((1, "per-query"), (2, "grouped")):
            check(lib().ndbhip_set_scan_mode(mode))
            for _ in range(2):
                ix.search_device(q[:nq], ot, od, oc, 1, nprobe, k, 0)
            check(lib().ndbhip_synchronize())
            reps = 10 if nq <= 256 else 4
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_device(q[:nq], ot, od, oc, 1, nprobe, k, 0)
            check(lib().ndbhip_synchronize())
            dt = (time.perf_counter() - t0) / reps
            line += f"   {name}: {dt * 1e3:8.3f} ms ({nq / dt:9.0f} q/s)"
        print(line)
    check(lib().ndbhip_set_scan_mode(0))


if __name__ == "__main__":
    main()
