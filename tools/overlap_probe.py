#!/usr/bin/env python3
"""A/B for VERDICT r4 item 1(c): does the NEXT batch's chain (centroid matrix, selection, seeds, pair tables, q - c
planes) run under THIS batch's sweep when the two batches are in flight at once?

The library keeps one stream and one set of per-batch arrays per process image, so the probe loads the shared object
TWICE (a byte copy under another name: two independent sets of globals), gives each copy its own HIP stream and its
own mirror of the same table, and drives them from two host threads (ctypes releases the GIL during a call):

  serial      one copy answers all the batches, one after the other (what bench.py times)
  overlapped  the two copies answer alternate batches concurrently
  threads     ONE copy of the library, two mirrors, two host threads that each gave themselves a stream
              (ndbhip_set_thread_stream): what bench.py --inflight 2 does

Same table, same queries, same results either way (checked); the only thing that changes is whether kernels of two
batches may share the device.  Prints queries/s of both and the kernels' own time per batch.

  python3 tools/overlap_probe.py [nvec dim lists probes batch steps]"""
import ctypes as C
import os
import shutil
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def load(path):
    L = C.CDLL(path)
    vp, i, i64 = C.c_void_p, C.c_int, C.c_int64
    L.ndbhip_init.argtypes = [i]
    L.ndbhip_set_stream.argtypes = [vp]
    L.ndbhip_set_thread_stream.argtypes = [vp]
    L.ndbhip_ivf_create.argtypes = [i, i, C.POINTER(vp)]
    L.ndbhip_ivf_build_device.argtypes = [vp, vp, vp, i64, i, C.POINTER(i)]
    L.ndbhip_ivf_prepare.argtypes = [vp, i]
    L.ndbhip_ivf_search_device.argtypes = [vp, vp, i, i, i, i, i64, vp, vp, vp]
    L.ndbhip_gen_rows_device.argtypes = [i, C.c_uint64, C.c_uint64, i64, i64, i, i, C.c_float, vp]
    L.ndbhip_synchronize.argtypes = []
    L.ndbhip_set_option.argtypes = [C.c_char_p, i]
    L.ndbhip_last_error.restype = C.c_char_p
    return L


def check(L, rc):
    if rc != 0:
        raise RuntimeError(f"ndbhip error {rc}: {L.ndbhip_last_error().decode()}")


def main():
    a = [int(x) for x in sys.argv[1:7]] + [None] * 6
    n, dim, nlists, nprobe, nq, steps = (a[0] or 1_000_000, a[1] or 768, a[2] or 1024, a[3] or 32, a[4] or 4096, a[5] or 40)
    k = 10
    src = os.environ.get("NDBHIP_LIB") or os.path.join(ROOT, "neurondb_amd", "lib", "libndbhip.so")
    twin = "/tmp/libndbhip_twin.so"
    shutil.copyfile(src, twin)
    libs = [load(src), load(twin)]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    opts = [o.split("=") for o in os.environ.get("NDB_OPTS", "").split(",") if o]
    for L, s in zip(libs, streams):
        check(L, L.ndbhip_init(0))
        check(L, L.ndbhip_set_stream(C.c_void_p(s.cuda_stream)))
        for name, val in opts:
            check(L, L.ndbhip_set_option(name.encode(), int(val)))
    L0 = libs[0]
    with torch.cuda.stream(streams[0]):
        base = torch.empty((n, dim), dtype=torch.float32, device=dev)
        check(L0, L0.ndbhip_gen_rows_device(1, 0x5EED0001, 0x5EEDC0DE, 0, n, dim, 1024, 0.1, C.c_void_p(base.data_ptr())))
        queries = torch.empty((nq * (steps + 4), dim), dtype=torch.float32, device=dev)
        check(L0, L0.ndbhip_gen_rows_device(1, 0x5EED0002, 0x5EEDC0DE, 0, queries.shape[0], dim, 1024, 0.1,
                                            C.c_void_p(queries.data_ptr())))
        rows = torch.arange(n, device=dev)
        blk = rows // 64
        tids = (((blk >> 16) & 0xFFFF) | ((blk & 0xFFFF) << 16) | ((rows % 64 + 1) << 32)).to(torch.int64)
    check(L0, L0.ndbhip_synchronize())
    torch.cuda.synchronize()
    handles = []
    for L, s in zip(libs, streams):
        h = C.c_void_p()
        it = C.c_int(0)
        check(L, L.ndbhip_ivf_create(dim, nlists, C.byref(h)))
        check(L, L.ndbhip_ivf_build_device(h, C.c_void_p(base.data_ptr()), C.c_void_p(tids.data_ptr()), n, 50, C.byref(it)))
        check(L, L.ndbhip_ivf_prepare(h, 1))
        check(L, L.ndbhip_synchronize())
        handles.append(h)
    del base
    outs = [(torch.zeros((steps + 4, nq, k), dtype=torch.int64, device=dev), torch.zeros((steps + 4, nq, k), dtype=torch.float32, device=dev),
             torch.zeros((steps + 4, nq), dtype=torch.int32, device=dev)) for _ in range(2)]

    def run(which, batches, res):
        L, h = libs[which], handles[which]
        ot, od, oc = outs[res]
        for b in batches:
            q = queries[b * nq:(b + 1) * nq]
            check(L, L.ndbhip_ivf_search_device(h, C.c_void_p(q.data_ptr()), nq, 1, nprobe, k, 0, C.c_void_p(ot[b].data_ptr()),
                                                C.c_void_p(od[b].data_ptr()), C.c_void_p(oc[b].data_ptr())))
        check(L, L.ndbhip_synchronize())

    for w in (0, 1):                       # warm-up of both copies (their first batch lays nothing out: prepared above)
        run(w, [steps, steps + 1, steps + 2, steps + 3], w)
    torch.cuda.synchronize()
    all_b = list(range(steps))
    t0 = time.perf_counter()
    run(0, all_b, 0)
    t_serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w, all_b[w::2], 1)) for w in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_over = time.perf_counter() - t0
    torch.cuda.synchronize()
    same = all(bool(torch.equal(outs[0][j][:steps], outs[1][j][:steps])) for j in range(3))
    # one copy of the library, a second mirror of its own, per-thread streams
    L = libs[0]
    h2 = C.c_void_p()
    it = C.c_int(0)
    with torch.cuda.stream(streams[0]):
        base = torch.empty((n, dim), dtype=torch.float32, device=dev)
        check(L, L.ndbhip_gen_rows_device(1, 0x5EED0001, 0x5EEDC0DE, 0, n, dim, 1024, 0.1, C.c_void_p(base.data_ptr())))
    check(L, L.ndbhip_ivf_create(dim, nlists, C.byref(h2)))
    check(L, L.ndbhip_ivf_build_device(h2, C.c_void_p(base.data_ptr()), C.c_void_p(tids.data_ptr()), n, 50, C.byref(it)))
    check(L, L.ndbhip_ivf_prepare(h2, 1))
    check(L, L.ndbhip_synchronize())
    del base
    hs = [handles[0], h2]
    s3 = [torch.cuda.Stream(), torch.cuda.Stream()]
    out3 = (torch.zeros_like(outs[0][0]), torch.zeros_like(outs[0][1]), torch.zeros_like(outs[0][2]))

    def run_t(which, batches):
        check(L, L.ndbhip_set_thread_stream(C.c_void_p(s3[which].cuda_stream)))
        for b in batches:
            q = queries[b * nq:(b + 1) * nq]
            check(L, L.ndbhip_ivf_search_device(hs[which], C.c_void_p(q.data_ptr()), nq, 1, nprobe, k, 0, C.c_void_p(out3[0][b].data_ptr()),
                                                C.c_void_p(out3[1][b].data_ptr()), C.c_void_p(out3[2][b].data_ptr())))
        check(L, L.ndbhip_synchronize())
        check(L, L.ndbhip_set_thread_stream(None))

    th = [threading.Thread(target=run_t, args=(w, [steps + w, steps + 2 + w])) for w in (0, 1)]      # warm-up
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run_t, args=(w, all_b[w::2])) for w in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    t_thr = time.perf_counter() - t0
    torch.cuda.synchronize()
    same3 = all(bool(torch.equal(outs[0][j][:steps], out3[j][:steps])) for j in range(3))
    print(f"overlap_probe: {n} x {dim}, lists {nlists}, probes {nprobe}, k {k}, {steps} batches of {nq} queries"
          f"{' (' + os.environ['NDB_OPTS'] + ')' if opts else ''}")
    print(f"  serial      {t_serial / steps * 1e3:.3f} ms per batch   {nq * steps / t_serial / 1e6:.3f} M q/s")
    print(f"  overlapped  {t_over / steps * 1e3:.3f} ms per batch   {nq * steps / t_over / 1e6:.3f} M q/s   "
          f"(two batches in flight: two copies of the library, two streams, two host threads)")
    print(f"  threads     {t_thr / steps * 1e3:.3f} ms per batch   {nq * steps / t_thr / 1e6:.3f} M q/s   "
          f"(one copy of the library, two mirrors, per-thread streams)")
    print(f"  ratio {t_serial / t_over:.3f} / {t_serial / t_thr:.3f}; results identical: {same} / {same3}")


if __name__ == "__main__":
    main()
