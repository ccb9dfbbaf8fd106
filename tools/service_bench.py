#!/usr/bin/env python3
"""One device-owner process + N backend processes over the shared-memory ring (include/ndb_service.h).

  owner   : builds / loads the index, ndbhip_init, ndb_service_serve_ivf  (the only process that touches the device)
  backend : ndb_client_connect, then one query at a time like ivfrescan + ivfgettuple would
            (`inflight` > 1 emulates that many backends per OS process: PostgreSQL installations run hundreds of
            connections, a test box does not want hundreds of Python interpreters)

Prints aggregate queries/s, the batch sizes the owner saw, and checks a sample of every backend's answers against
the CPU oracle (ids + float4 bits).  usage: tools/service_bench.py [--backends 16] [--inflight 1] [--queries 2000] ..."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def make_index_arrays(n, dim, nlists, seed):
    from tests.util import make_ivf_arrays
    return make_ivf_arrays(n, dim, nlists, seed=seed, dup_frac=0.02)


def c2_queries(rank, nq, dim, components):
    """`nq` queries of backend `rank` from the bench's generator (clustered: the table's own components), made on the
    host: ndbhip_gen_rows_host is plain C and returns the bits bench.py's device generator returns"""
    from neurondb_amd import _lib
    q = np.zeros((nq, dim), np.float32)
    _lib.check(_lib.lib().ndbhip_gen_rows_host(1, 0x5EED0002, 0x5EEDC0DE, rank * nq, nq, dim, components, C.c_float(0.1),
                                               q.ctypes.data_as(C.c_void_p)))
    return q


def backend(rank, name, dim, nq, inflight, nprobe, k, seed, ret, barrier, c2_components=0):
    """A PostgreSQL backend: no device, no index — only the ring."""
    from neurondb_amd import _lib
    L = _lib.lib()
    c = C.c_void_p()
    for _ in range(200):                                   # the owner may still be loading its index
        if L.ndb_client_connect(name.encode(), C.byref(c)) == 0:
            break
        time.sleep(0.05)
    else:
        ret[rank] = "connect failed"
        return
    rng = np.random.default_rng(seed + rank)
    q = c2_queries(rank, nq, dim, c2_components) if c2_components else rng.standard_normal((nq, dim)).astype(np.float32)
    tids = np.zeros((nq, k, 6), np.uint8)
    dist = np.zeros((nq, k), np.float32)
    cnt = np.zeros(nq, np.int32)
    barrier.wait()
    t0 = time.perf_counter()
    tickets, nxt, done = [], 0, 0
    one = C.c_int()
    while done < nq:
        while nxt < nq and len(tickets) < inflight:
            t = C.c_int()
            _lib.check(L.ndb_client_submit(c, q[nxt].ctypes.data_as(C.c_void_p), 1, nprobe, k, 0, C.byref(t)))
            tickets.append((t.value, nxt))
            nxt += 1
        tk, i = tickets.pop(0)
        _lib.check(L.ndb_client_wait(c, tk, tids[i].ctypes.data_as(C.c_void_p), dist[i].ctypes.data_as(C.c_void_p),
                                     C.byref(one), 20000))
        cnt[i] = one.value
        done += 1
    wall = time.perf_counter() - t0
    L.ndb_client_disconnect(c)
    ret[rank] = dict(wall=wall, q=q, tids=tids, dist=dist, cnt=cnt)


def owner_gpu(name, arrays, nslots, max_batch, linger_us, ready, stats_out):
    from neurondb_amd import IvfIndex, _lib
    L = _lib.lib()
    _lib.ensure_init(0)
    if "c2" in arrays:
        # the headline table (bench.py's generator and build rule), built on the device; the image goes to /dev/shm for
        # the parent's oracle check
        import torch
        from bench import make_data, pack_tids
        n, dim, nlists = arrays["c2"]
        dev = torch.device("cuda", 0)
        _lib.use_torch_stream()
        base = make_data(n, dim, "clustered", nlists, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
        ix = IvfIndex(dim, nlists)
        ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
        ix.prepare(1)
        del base
        cent, ll, rows, tids = ix.export(rows=True)
        np.save(arrays["image"] + "_cent.npy", cent)
        np.save(arrays["image"] + "_ll.npy", ll)
        np.save(arrays["image"] + "_rows.npy", rows)
        np.save(arrays["image"] + "_tids.npy", np.ascontiguousarray(tids).view(np.uint8))
        del rows, tids
        sdim = dim
    else:
        ix = IvfIndex(arrays["centroids"].shape[1], len(arrays["list_len"]))
        ix.set_centroids(arrays["centroids"])
        ix.load(arrays["list_len"], arrays["rows"], arrays["tids"])
        sdim = arrays["centroids"].shape[1]
    s = C.c_void_p()
    _lib.check(L.ndb_service_create(name.encode(), sdim, 64, nslots, C.byref(s)))
    ready.set()
    st = _lib.ServiceStats()
    _lib.check(L.ndb_service_serve_ivf(s, ix._h, max_batch, linger_us, 0, C.byref(st)))
    stats_out.update(batches=st.batches, queries=st.queries, max_batch=st.max_batch, busy_s=st.busy_s)
    L.ndb_service_destroy(s)
    ix.close()


def run(backends=16, inflight=1, queries=500, n=20000, dim=128, nlists=64, nprobe=8, k=10, nslots=2048,
        max_batch=4096, linger_us=100, check=64, seed=5, data="random", clients="python"):
    """data "random": a random test index and N(0,1) queries that have nothing to do with it (every list is as far as any
    other: the scan excludes little); "c2": bench.py's headline table and queries drawn like its rows"""
    from oracle import ndbo
    from tests.util import oracle_image
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret, stats = mgr.dict(), mgr.dict()
    name = f"/ndb_service_bench_{os.getpid()}"
    image = f"/dev/shm/ndb_service_bench_{os.getpid()}"
    arrays = {"c2": (n, dim, nlists), "image": image} if data == "c2" else make_index_arrays(n, dim, nlists, seed)
    ready = ctx.Event()
    barrier = ctx.Barrier(backends)
    own = ctx.Process(target=owner_gpu, args=(name, arrays, nslots, max_batch, linger_us, ready, stats))
    own.start()
    if not ready.wait(300):
        own.terminate()
        raise RuntimeError("the owner did not come up")
    if clients == "c":
        # the backends as threads of one plain-C process (examples/service_clients.c): the ring without an interpreter
        # between a wake-up and the next request
        import subprocess
        if data != "c2":
            raise SystemExit("--clients c draws its queries from the bench generator: use --data c2")
        dump = image + "_answers.bin"
        exe = os.path.join(ROOT, "neurondb_amd", "lib", "service_clients")
        out = subprocess.run([exe, name, str(backends), str(inflight), str(queries), str(nprobe), str(k), str(dim),
                              str(nlists), str(min(check, queries)), dump], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError(out.stdout + out.stderr)
        cj = json.loads(out.stdout.strip().splitlines()[-1])
        ck = min(check, queries)
        raw = np.fromfile(dump, dtype=np.uint8)
        os.unlink(dump)
        per = ck * dim * 4 + ck * 4 + ck * k * 6 + ck * k * 4
        for r in range(backends):
            blk = raw[r * per:(r + 1) * per]
            o = 0
            qv = blk[o:o + ck * dim * 4].view(np.float32).reshape(ck, dim); o += ck * dim * 4
            cn = blk[o:o + ck * 4].view(np.int32); o += ck * 4
            td = blk[o:o + ck * k * 6].reshape(ck, k, 6); o += ck * k * 6
            ds = blk[o:o + ck * k * 4].view(np.float32).reshape(ck, k)
            ret[r] = dict(wall=cj["wall_s"], q=qv.copy(), tids=td.copy(), dist=ds.copy(), cnt=cn.copy())
    else:
        procs = [ctx.Process(target=backend, args=(r, name, dim, queries, inflight, nprobe, k, 1000 * seed, ret, barrier,
                                                   nlists if data == "c2" else 0))
                 for r in range(backends)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
    # stop the owner through the ring, like a shutting-down postmaster would
    from neurondb_amd import _lib
    L = _lib.lib()
    c = C.c_void_p()
    _lib.check(L.ndb_client_connect(name.encode(), C.byref(c)))
    L.ndb_client_stop_service(c)
    L.ndb_client_disconnect(c)
    own.join(60)
    bad = [r for r in range(backends) if not isinstance(ret.get(r), dict)]
    if bad:
        raise RuntimeError({r: ret.get(r) for r in bad})
    wall = max(ret[r]["wall"] for r in range(backends))
    if data == "c2":
        ll = np.load(image + "_ll.npy")
        off = np.zeros(len(ll) + 1, np.int64)
        off[1:] = np.cumsum(ll)
        img = ndbo.IvfImage(np.load(image + "_cent.npy"), off, np.load(image + "_rows.npy"),
                            np.load(image + "_tids.npy").view(ndbo.TID_DTYPE).reshape(-1))
        for suffix in ("_cent.npy", "_ll.npy", "_rows.npy", "_tids.npy"):
            os.unlink(image + suffix)
    else:
        img = oracle_image(arrays)
    mism = 0
    for r in range(backends):
        d = ret[r]
        for i in range(min(check, queries)):
            et, ed, _ = img.search(d["q"][i], 1, nprobe, k, 0)
            got_t = d["tids"][i, :d["cnt"][i]].copy().view(ndbo.TID_DTYPE).reshape(-1)
            mism += not (d["cnt"][i] == len(et) and np.array_equal(ndbo.tids_to_u64(got_t), ndbo.tids_to_u64(et)) and
                         np.array_equal(d["dist"][i, :len(et)].view(np.uint32), ed.view(np.uint32)))
    return {"backends": backends, "inflight_per_backend": inflight, "queries": backends * queries,
            "aggregate_queries_per_s": round(backends * queries / wall, 1), "wall_s": round(wall, 3),
            "owner": dict(stats), "avg_batch": round(stats.get("queries", 0) / max(1, stats.get("batches", 1)), 1),
            "checked_against_oracle": backends * min(check, queries), "mismatches": int(mism),
            "index": f"{n}x{dim} lists={nlists} probes={nprobe} k={k}",
            "clients": "one Python process per backend" if clients == "python" else "threads of examples/service_clients.c",
            "data": "bench.py's clustered table and query stream (C2)" if data == "c2" else
                    "random test index, N(0,1) queries unrelated to it"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    for a, d in (("backends", 16), ("inflight", 1), ("queries", 500), ("n", 20000), ("dim", 128), ("nlists", 64),
                 ("nprobe", 8), ("k", 10), ("nslots", 2048), ("max_batch", 4096), ("linger_us", 100), ("check", 64)):
        ap.add_argument("--" + a.replace("_", "-"), type=int, default=d)
    ap.add_argument("--data", choices=["random", "c2"], default="random")
    ap.add_argument("--clients", choices=["python", "c"], default="python")
    print(json.dumps(run(**vars(ap.parse_args()))))
