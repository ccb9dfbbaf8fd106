#!/usr/bin/env python3
"""The centred sweep on the i.i.d. N(0,1) table (BASELINE.md's C2 data: 1M x 768, lists 1024, probes 32, k 10,
4096 queries per step) under a list of option settings, one table build for all of them.
usage: tools/dense_probe.py "name=value,name=value" "name=value" ...   (each argument is one variant; "" = defaults)
Prints per variant: ms per step, ms per sweep launch (the library's HIP events), issued TFLOP/s, and — unless the
variant sets screen16_debug — whether the results equal the first variant's (TIDs and float4 bits)."""
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
    _lib.use_torch_stream()
    n, dim = int(os.environ.get("NVEC", 1_000_000)), int(os.environ.get("DIM", 768))
    kind = os.environ.get("DATA", "gauss")
    nq = int(os.environ.get("NQ", 4096))
    steps = int(os.environ.get("STEPS", 6))
    sigma = float(os.environ.get("SIGMA", 0.1))
    base = make_data(n, dim, kind, 1024, sigma, 0x5EED0001, 0x5EEDC0DE, dev)
    qd = make_data(nq * 4, dim, kind, 1024, sigma, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, int(os.environ.get("LISTS", 1024)))
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    ot = torch.zeros((nq, 10), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, 10), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    check(lib().ndbhip_profile(1))
    ref = None
    variants = sys.argv[1:] or [""]
    defaults = {}
    default_of = {"screen16c_epi": 1, "screen16c_rot": 0, "screen16c_dense": 1, "screen16c_pfd": 0, "screen16c_sample": 2048, "screen16c_tight": 128, "screen16c_dense_split": 3, "screen16c_dense_sync": 16, "screen16c_dense_small": 1}
    for v in variants:
        opts = dict(kv.split("=") for kv in v.split(",") if kv)
        for k in list(defaults):
            if k not in opts:
                check(lib().ndbhip_set_option(k.encode(), defaults.pop(k)))
        for k, val in opts.items():
            defaults.setdefault(k, default_of.get(k, 0))
            check(lib().ndbhip_set_option(k.encode(), int(val)))
        for i in range(2):
            ix.search_device(qd[(i % 4) * nq:(i % 4 + 1) * nq], ot, od, oc, 1, 32, 10, 0)
        check(lib().ndbhip_synchronize())
        s0 = _lib.stats()
        t0 = time.perf_counter()
        for i in range(steps):
            ix.search_device(qd[(i % 4) * nq:(i % 4 + 1) * nq], ot, od, oc, 1, 32, 10, 0)
        check(lib().ndbhip_synchronize())
        dt = (time.perf_counter() - t0) / steps
        s1 = _lib.stats()
        launches = max(1, s1["scan_launches"] - s0["scan_launches"])
        kms = (s1["scan_kernel_ms"] - s0["scan_kernel_ms"]) / launches
        swept = (s1.get("rows_swept", 0) - s0.get("rows_swept", 0)) / max(1, steps)
        tf = swept * 2 * ((dim + 63) // 64 * 64) / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        # the results of the LAST step's slice, compared across variants
        ix.search_device(qd[:nq], ot, od, oc, 1, 32, 10, 0)
        check(lib().ndbhip_synchronize())
        res = (ot.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy(), oc.cpu().numpy().copy())
        same = ""
        if "screen16_debug" not in opts:
            if ref is None:
                ref = res
                same = "(reference)"
            else:
                same = "same results" if all(np.array_equal(a, b) for a, b in zip(ref, res)) else "RESULTS DIFFER"
        fb = s1["screen16_fallbacks"] - s0["screen16_fallbacks"]
        em = (s1["rows_emitted"] - s0["rows_emitted"]) / steps / nq
        rs = (s1["rows_rescored"] - s0["rows_rescored"]) / steps / nq
        import ctypes as _C
        ph = (_C.c_ulonglong * 64)()
        check(lib().ndbhip_debug_phases(ph))
        if ph[32 + 6]:
            # a profiling build (make EXTRA=-DNDB_PHASES): block 0's wave 0 (a loader) and wave 4 (a multiplier), the last launch
            names = ("DMA wait", "barrier", "request", "multiply", "results", "tighten")
            for w, who in ((0, "loader"), (1, "multiplier")):
                items = max(1, ph[32 + 8 * w + 6])
                print("    " + who + ", us per item: " + ", ".join(f"{nm} {ph[32 + 8 * w + i] / 100 / items:.2f}" for i, nm in enumerate(names)) +
                      f"; {items} items", flush=True)
        print(f"{v or 'defaults':48s} step {dt * 1e3:8.3f} ms  sweep {kms:8.3f} ms  issued {tf:7.1f} TFLOP/s  "
              f"launches/step {launches / steps:.1f} fallbacks {fb} emitted/q {em:.0f} rescored/q {rs:.1f} {same}", flush=True)


if __name__ == "__main__":
    main()
