#!/usr/bin/env python3
"""Replays a case tools/fuzz_scan.py saved on a mismatch (gpurun_out/fuzz_fail_*.npz): the same mirror, queries and options
through all five scan modes against the oracle.   usage: tools/fuzz_replay.py FILE.npz [name=value ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from tests.util import assert_same_results, oracle_image, oracle_search_batch


def load_case(path):
    """(file, mirror, queries, k, nprobe, cap, strategy, options) of a saved case"""
    from neurondb_amd import IvfIndex
    z = np.load(path)
    a = dict(centroids=z["centroids"], list_len=z["list_len"], rows=z["rows"], tids=z["tids"])
    dim, nlists = a["centroids"].shape[1], len(a["list_len"])
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(a["centroids"])
    if bool(z["half"]):
        ix.load_f16(a["list_len"], z["rows_f16"], a["tids"])
    else:
        ix.load(a["list_len"], a["rows"], a["tids"])
    opts = dict(zip([str(n) for n in z["opt_names"]], [int(v) for v in z["opt_values"]]))
    return z, ix, z["q"], int(z["k"]), int(z["nprobe"]), int(z["cap"]), int(z["strategy"]), opts


def main():
    from neurondb_amd import _lib
    from neurondb_amd._lib import check, lib
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    z, ix, q, k, nprobe, cap, strategy, opts = load_case(sys.argv[1])
    a = dict(centroids=z["centroids"], list_len=z["list_len"], rows=z["rows"], tids=z["tids"])
    dim, nlists = a["centroids"].shape[1], len(a["list_len"])
    for kv in sys.argv[2:]:
        opts[kv.split("=")[0]] = int(kv.split("=")[1])
    print("case", dict(dim=dim, n=len(a["rows"]), nlists=nlists, nq=len(q), k=k, nprobe=nprobe, cap=cap, strategy=strategy,
                       failed_mode=int(z["mode"]), lens=a["list_len"].tolist()), "\nopts", opts, flush=True)
    for name, value in opts.items():
        check(lib().ndbhip_set_option(name.encode(), value))
    et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k, cap)
    for rep in range(2):
        for mode in (5, 3, 2, 1, 0):
            check(lib().ndbhip_set_scan_mode(mode))
            t, d, c = ix.search(q, strategy, nprobe, k, cap)
            try:
                assert_same_results(t, d, c, et, ed, ec)
                print(" mode", mode, "same", flush=True)
            except AssertionError as e:
                print(" mode", mode, "MISMATCH", str(e)[:300], flush=True)


if __name__ == "__main__":
    main()
