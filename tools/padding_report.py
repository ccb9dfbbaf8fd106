#!/usr/bin/env python3
"""How much of the grouped scan's arithmetic is padding: a work item scores a 64-row tile against a group of
16 queries, so a list probed by cnt queries and holding len rows costs ceil(cnt/16)*16 x ceil(len/64)*64
(row, query) sums for cnt x len useful ones.  Reports the ratio for bench.py's default workload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import make_data, pack_tids


def main():
    from neurondb_amd import IvfIndex, _lib
    dev = torch.device("cuda", 0)
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    n, dim, nlists, nprobe = 1_000_000, 768, 1024, 32
    base = make_data(n, dim, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, nlists)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    _, ll, _, _ = ix.export(rows=False)
    for nq in (64, 1024, 2048, 4096):
        q = make_data(nq, dim, "clustered", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev).cpu().numpy()
        probes = ix.select_clusters(q, nprobe)
        cnt = np.bincount(probes[probes >= 0].ravel(), minlength=nlists).astype(np.int64)
        useful = int((cnt * ll).sum())
        done = int((((cnt + 15) // 16) * 16 * ((ll + 63) // 64) * 64).sum())
        by_q = int((((cnt + 15) // 16) * 16 * ll).sum())
        print(f"nq={nq:5d}: useful {useful:.3e} pair-rows, computed {done:.3e} (x{done / useful:.3f}); "
              f"query-group padding alone x{by_q / useful:.3f}, row-tile padding alone x{done / by_q:.3f}")


if __name__ == "__main__":
    main()
