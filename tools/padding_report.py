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
        for qt, rt in ((128, 128), (64, 128), (128, 256), (64, 256), (32, 128), (256, 128)):
            d2 = int((((cnt + qt - 1) // qt) * qt * ((ll + rt - 1) // rt) * rt).sum())
            items = int((((cnt + qt - 1) // qt) * ((ll + rt - 1) // rt)).sum())
            print(f"          fp16 matrix-core sweep, tiles of {qt} queries x {rt} rows: x{d2 / useful:.3f} ({items} items)")
        # mixed: lists take 128-query tiles, the remainder of <= 64 queries a 64-query tile
        rem = cnt % 128
        full = (cnt // 128) * 128
        mixed = (full + np.where(rem == 0, 0, np.where(rem <= 64, 64, 128))) * ((ll + 127) // 128) * 128
        print(f"          128 x 128 tiles with a 64-query tile for remainders <= 64: x{int(mixed.sum()) / useful:.3f}")
        if nq == 4096:
            order = np.argsort(-ll)[:8]
            print("          longest lists (len, queries probing):", [(int(ll[i]), int(cnt[i])) for i in order])
            print("          cnt quantiles:", np.quantile(cnt, [0, .1, .25, .5, .75, .9, 1]).tolist(), "mean", cnt.mean())


if __name__ == "__main__":
    main()
