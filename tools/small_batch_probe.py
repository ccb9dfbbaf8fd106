#!/usr/bin/env python3
"""Host-pointer and device-pointer ndbhip_ivf_search latency for small batches (what a backend serving a few
concurrent scans sees), 1M x 768, lists = 1024, probes = 32, k = 10."""
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
    n, dim = int(os.environ.get("NVEC", 1_000_000)), 768
    base = make_data(n, dim, "clustered", 1024, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    qd = make_data(4096, dim, "clustered", 1024, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    q = qd.cpu().numpy()
    ix = IvfIndex(dim, 1024)
    ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    if os.environ.get("SCAN_MODE"):
        check(lib().ndbhip_set_scan_mode(int(os.environ["SCAN_MODE"])))      # 1 per-query scan, 2 grouped scan
    for kv in [x for x in os.environ.get("OPTS", "").split(",") if x]:
        check(lib().ndbhip_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
        print("option", kv, flush=True)
    for minnq in [int(x) for x in os.environ.get("MINNQS", "0").split(",")]:
      if minnq > 0:                       # (0: the library's default crossover)
          check(lib().ndbhip_set_option(b"screen_min_nq", minnq))
          print("screen_min_nq", minnq, flush=True)
      for nq in [int(x) for x in os.environ.get("NQS", "1,2,4,7,8,9,16,32,64,128,256,512").split(",")]:
          ot = torch.zeros((nq, 10), dtype=torch.int64, device=dev)
          od = torch.zeros((nq, 10), dtype=torch.float32, device=dev)
          oc = torch.zeros(nq, dtype=torch.int32, device=dev)
          res = []
          for host in (True, False):
              def run(i):
                  if host:
                      ix.search(q[i * nq:(i + 1) * nq], 1, 32, 10)
                  else:
                      ix.search_device(qd[i * nq:(i + 1) * nq], ot, od, oc, 1, 32, 10, 0)
                      check(lib().ndbhip_synchronize())
              for i in range(3):
                  run(i)
              ts = []
              for i in range(3, 8):
                  t0 = time.perf_counter()
                  run(i % (4096 // nq))
                  ts.append(time.perf_counter() - t0)
              res.append(np.median(ts) * 1e3)
          print(f"nq={nq:4d}  host pointers {res[0]:7.3f} ms   device pointers {res[1]:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
