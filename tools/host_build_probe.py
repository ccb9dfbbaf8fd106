#!/usr/bin/env python3
"""ndbhip_ivf_build from pageable host memory (with its timeline: option debug_build = 2), next to torch's own
pageable upload of the same table (1M x 768 fp32, 3 GB)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_data, pack_tids


def main():
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    n, dim = int(os.environ.get("NVEC", 1_000_000)), 768
    base = make_data(n, dim, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    host = base.cpu().numpy()
    tids = pack_tids(torch.arange(n, device=dev)).cpu().numpy()
    del base
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = torch.from_numpy(host).cuda()
        torch.cuda.synchronize()
        print(f"torch pageable .cuda(): {time.perf_counter() - t0:.4f} s = {host.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
        del x
    for th in [0, 0]:
        for rep in range(3):
            check(lib().ndbhip_set_option(b"debug_build", 2 if rep == 2 else 0))
            ix = IvfIndex(dim, 1024)
            t0 = time.perf_counter()
            ix.build(host, tids, 50)
            t = time.perf_counter() - t0
            ix.close()
        print(f" ndbhip_ivf_build from host {t:.4f} s = {n / t / 1e6:.2f} M vec/s ({host.nbytes / t / 1e9:.1f} GB/s of table)", flush=True)


if __name__ == "__main__":
    main()
