#!/usr/bin/env python3
"""BASELINE configs[0] on the CPU oracle: 10k x 128 fp32, IVFFlat lists=100, k=10, L2.
Reports recall@10 of the reference-compatible mode (nprobe pinned to 10, candidate cap k*10 = 100:
quirks Q3/Q4) and of the intended mode (every entry of the probed lists scored), and single-thread
query latency.  Test/measurement infrastructure (uses oracle/)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import ndbo


def main():
    rng = np.random.default_rng(0x5EED0001)
    n, dim, nlists, k = 10000, 128, 100, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((200, dim)).astype(np.float32)
    t0 = time.perf_counter()
    img, asg, iters = ndbo.build_ivf_image(base, nlists)
    tb = time.perf_counter() - t0
    ll = np.diff(img.list_off)
    print(f"build (oracle, 1 thread): {tb:.1f} s, {iters} Lloyd iterations, list sizes min/mean/max "
          f"{ll.min()}/{ll.mean():.0f}/{ll.max()}")
    d2 = ((q[:, None, :].astype(np.float64) - base[None].astype(np.float64)) ** 2).sum(-1)
    gt = np.argsort(d2, axis=1, kind="stable")[:, :k]
    tid_of = {int(v): i for i, v in enumerate(ndbo.tids_to_u64(ndbo.tids_from_rows(np.arange(n))))}
    for name, nprobe, cap in (("ref_compat (nprobe=10, cap=100)", 10, 100), ("intended nprobe=10", 10, 0),
                              ("intended nprobe=32", 32, 0), ("intended nprobe=100 (all lists)", 100, 0)):
        rec, scored = [], 0
        t0 = time.perf_counter()
        for i, qq in enumerate(q):
            t, d, ns = img.search(qq, 1, nprobe, k, cap)
            scored += ns
            got = {tid_of[int(v)] for v in ndbo.tids_to_u64(t)}
            rec.append(len(got & set(gt[i].tolist())) / k)
        dt = (time.perf_counter() - t0) / len(q)
        print(f"{name:34s} recall@10 {np.mean(rec):.3f}   {scored / len(q):7.0f} entries scored/query   "
              f"{dt * 1e3:.2f} ms/query (1 thread)")


if __name__ == "__main__":
    main()
