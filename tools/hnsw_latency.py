#!/usr/bin/env python3
"""Single-query latency of ndbhip_hnsw_search through host pointers (what one hnswgettuple costs), 200k x 768."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import numpy as np
import torch

from bench import pack_tids


def main():
    from neurondb_amd import HnswIndex, _lib
    lib, check = _lib.lib(), _lib.check
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    n, dim, m, efc = int(os.environ.get("NVEC", 200000)), 768, 16, 200
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    base = torch.randn((n, dim), generator=g, device=dev)
    base = base / base.norm(dim=1, keepdim=True)
    q = torch.randn((400, dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)
    ix = HnswIndex(dim, m)
    check(lib.ndbhip_hnsw_build_device(ix._h, C.c_void_p(base.data_ptr()),
                                       C.c_void_p(pack_tids(torch.arange(n, device=dev)).data_ptr()), n,
                                       levels.ctypes.data, efc))
    check(lib.ndbhip_synchronize())
    for i in range(20):
        ix.search(q[i:i + 1], 2, 64, 10)
    ts = []
    for i in range(20, 320):
        t0 = time.perf_counter()
        ix.search(q[i:i + 1], 2, 64, 10)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"single-query ndbhip_hnsw_search latency: p50 {np.percentile(ts, 50):.0f} us, p90 {np.percentile(ts, 90):.0f} us, "
          f"p99 {np.percentile(ts, 99):.0f} us ({n}x{dim}, m={m}, ef_search=64, k=10, cosine)")


if __name__ == "__main__":
    main()
