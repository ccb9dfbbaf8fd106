#!/usr/bin/env python3
"""Randomised parity campaign: IVF search on the device (all scan modes, screened included) against the CPU
oracle over random shapes, data scales, k, nprobe, candidate caps and strategies.
usage: python tools/fuzz_scan.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import ndbo
from tests.util import assert_same_results, oracle_image, oracle_search_batch


_LUT = None


def sharded_case(rng, ix, img, q, strategy, nprobe, k, cap, et, ed, ec):
    """the same search through 2..4 slice shards + merge (what an N-GPU run does), screened mode"""
    import torch
    from neurondb_amd import _lib
    from neurondb_amd.dist import partition_slices
    world = int(rng.integers(2, 5))
    _, ll, _, _ = ix.export(rows=False)
    lo, ln, tail = partition_slices(ll, world, None, split_frac=0.0, align=int(rng.choice([16, 64])))
    shards = [ix.shard_slices(lo[w], ln[w], tail[w]) for w in range(world)]
    dq = torch.from_numpy(q).cuda()
    nq, rcap = len(q), 3 * k
    cand = torch.zeros((world, nq, rcap, 2), dtype=torch.int64, device="cuda")
    ncand = torch.zeros((world, nq), dtype=torch.int32, device="cuda")
    total = torch.zeros((world, nq), dtype=torch.int64, device="cuda")
    for w, sh in enumerate(shards):
        sh.search_partial_device(dq, cand[w], ncand[w], total[w], strategy, nprobe, k, cap)
    ot = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    od = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    oc = torch.zeros(nq, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ndbhip_merge_topk_device(cand.data_ptr(), ncand.data_ptr(), total[0].data_ptr(), world, nq, k,
                                                   rcap, ot.data_ptr(), od.data_ptr(), oc.data_ptr()))
    _lib.check(_lib.lib().ndbhip_synchronize())
    c = oc.cpu().numpy()
    t = ndbo.tids_from_device_u64(ot.cpu().numpy())
    d = od.cpu().numpy()
    assert np.array_equal(c, ec)
    for i in range(nq):
        assert np.array_equal(t[i, :c[i]], et[i, :c[i]]) and \
            np.array_equal(d[i, :c[i]].view(np.uint32), ed[i, :c[i]].view(np.uint32)), ("sharded", world, i)
    for sh in shards:
        sh.close()


# every option one_case draws, at the library's default: a campaign must leave the process as it found it (the `-m gpu`
# suite runs a few seconds of this in the middle of everything else)
DEFAULT_OPTIONS = {"screen16_sub_min": 256, "screen16_sub_rows": 128, "screen16_sublists": 1, "screen16_prune": 1,
                   "screen16_tighten": 1, "screen16c_qb": 0, "screen16c_dense": 1, "screen16c_sample": 2048,
                   "screen16c_tight": 128, "screen16_ip_centered": 1, "screen16_stage": 1, "screen16c_wave": 2,
                   "screen16c_wave_blocks": 2, "screen16c_wave_min_nq": 1024, "screen16c_plane_seeds": 1, "screen16c_bigk": 1}


def reset_options(lib, check):
    for name, value in DEFAULT_OPTIONS.items():
        check(lib.ndbhip_set_option(name.encode(), value))
    check(lib.ndbhip_set_scan_mode(0))


def one_case(rng, lib, IvfIndex, check):
    dim = int(rng.choice([64, 128, 192, 256, 768]))
    n = int(rng.integers(200, 5000))
    nlists = int(rng.integers(1, 24))
    nq = int(rng.choice([5, 16, 17, 64, 130, 200]))
    kind = rng.choice(["normal", "offset", "scaled", "integer", "clustered"])
    if kind == "integer":
        base = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
        q = rng.integers(-3, 4, size=(nq, dim)).astype(np.float32)
    elif kind == "offset":
        c = rng.standard_normal(dim).astype(np.float32) * float(10.0 ** rng.uniform(0, 3))
        s = float(10.0 ** rng.uniform(-3, 0))
        base = (c + s * rng.standard_normal((n, dim))).astype(np.float32)
        q = (c + s * rng.standard_normal((nq, dim))).astype(np.float32)
    elif kind == "scaled":
        base = (rng.standard_normal((n, dim)) * 10.0 ** rng.uniform(-4, 4, (n, 1))).astype(np.float32)
        q = (rng.standard_normal((nq, dim)) * 10.0 ** rng.uniform(-4, 4, (nq, 1))).astype(np.float32)
    elif kind == "clustered":
        cen = rng.standard_normal((8, dim)).astype(np.float32)
        base = (cen[rng.integers(0, 8, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
        q = (cen[rng.integers(0, 8, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
    if rng.random() < 0.3:
        base[rng.integers(0, n, n // 10)] = base[rng.integers(0, n, n // 10)]       # duplicates
    if rng.random() < 0.2:
        q[0] = base[0]
    if rng.random() < 0.25:
        # zero vectors: the reference's cosine distance of one is exactly 1, its inner product 0 — and a bound that took
        # it for a unit vector would be wrong (round 3's centred cosine did, once)
        base[rng.integers(0, n, 3)] = 0.0
        if rng.random() < 0.5:
            q[nq - 1] = 0.0
    cent = base[rng.choice(n, nlists, replace=False)].copy()
    asg = rng.integers(0, nlists, n) if rng.random() < 0.3 else \
        (((base[:, None, :64].astype(np.float64) - cent[None, :, :64]) ** 2).sum(-1)).argmin(1)
    order = np.argsort(asg, kind="stable")
    a = dict(centroids=cent, list_len=np.bincount(asg, minlength=nlists).astype(np.int64),
             rows=np.ascontiguousarray(base[order]), tids=ndbo.tids_from_rows(order))
    ix = IvfIndex(dim, nlists)
    ix.set_centroids(a["centroids"])
    half = kind in ("normal", "clustered", "integer") and rng.random() < 0.35
    if half:
        # a halfvec column: rows kept as fp16 (some of them subnormal); the oracle sees what fp16_to_float gives
        h = (a["rows"] * np.float32(rng.choice([1.0, 1e-3]))).astype(np.float16).view(np.uint16).copy()
        global _LUT
        if _LUT is None:
            _LUT = np.array([ndbo.lib().ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
        a = dict(a, rows=_LUT[h])
        ix.load_f16(a["list_len"], h, a["tids"])
        kind = kind + "/fp16"
    else:
        ix.load(a["list_len"], a["rows"], a["tids"])
    img = oracle_image(a)
    k = int(rng.choice([1, 10, 37, 100, 65, 160, 256]))     # (64 < k <= 256: the fp16 screen's radius thresholds, round 5)
    nprobe = int(rng.integers(1, nlists + 3))
    cap = int(rng.choice([0, 0, k * 10, 500]))
    strategy = int(rng.choice([1, 1, 1, 2, 3]))
    et, ed, ec, _ = oracle_search_batch(img, q, strategy, nprobe, k, cap)
    # the matrix-core screen's own variants: lists regrouped into sublists from 256 rows up (so that these small
    # tables have some), 32- to 256-row sublists, list-level exclusion and in-sweep tightening on or off
    sub_opts = [int(rng.choice([256, 256, 2048])), int(rng.choice([32, 128, 256])), int(rng.random() < 0.8),
                int(rng.random() < 0.8), int(rng.random() < 0.8),
                # the centred sweep's tile: chosen by the library, or forced to 32 / 128 pairs x 128 rows, or 256 x 256
                int(rng.choice([0, 0, 1, 4, 8, 8]))]
    for name, value in zip(("screen16_sub_min", "screen16_sub_rows", "screen16_sublists", "screen16_prune", "screen16_tighten",
                            "screen16c_qb"), sub_opts):
        check(lib.ndbhip_set_option(name.encode(), value))
    # round 4: the dense tile's own kernel or the older 8-wave sweep, sample-seeded thresholds on or off, how often the
    # dense kernel tightens, inner product on the centred planes or on two planes, rows streamed through LDS (ring depth)
    r4 = {"screen16c_dense": int(rng.random() < 0.75), "screen16c_sample": int(rng.choice([0, 256, 2048])),
          "screen16c_tight": int(rng.choice([8, 128, 1024])), "screen16_ip_centered": int(rng.random() < 0.75),
          "screen16_stage": int(rng.choice([0, 1, 1, 2, 5, 13])),
          # round 5: the 32-pair tile as wave-autonomous register streams (chunks in flight per wave) or the LDS ring
          "screen16c_wave": int(rng.choice([0, 2, 2, 3, 4])),
          "screen16c_wave_blocks": int(rng.choice([1, 2, 3, 3])), "screen16c_wave_min_nq": 1,
          "screen16c_plane_seeds": int(rng.random() < 0.7), "screen16c_bigk": int(rng.random() < 0.85)}
    for name, value in r4.items():
        check(lib.ndbhip_set_option(name.encode(), value))
    if os.environ.get("FUZZ_TRACE"):
        print("OPTS", r4, flush=True)
    if os.environ.get("FUZZ_TRACE"):
        print("CASE", dict(dim=dim, n=n, nlists=nlists, nq=nq, kind=kind, k=k, nprobe=nprobe, cap=cap, strategy=strategy,
                           lens=a["list_len"].tolist()), flush=True)
    for mode in (5, 3, 2, 1, 0):
        check(lib.ndbhip_set_scan_mode(mode))
        if os.environ.get("FUZZ_TRACE"):
            print(" mode", mode, flush=True)
        t, d, c = ix.search(q, strategy, nprobe, k, cap)
        try:
            assert_same_results(t, d, c, et, ed, ec)
        except AssertionError:
            print("MISMATCH", dict(dim=dim, n=n, nlists=nlists, nq=nq, kind=kind, k=k, nprobe=nprobe, cap=cap,
                                   strategy=strategy, mode=mode), flush=True)
            # the case itself, for tools/fuzz_replay.py (gpurun_out/ comes back from the GPU box)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            opts = dict(r4, screen16_sub_min=sub_opts[0], screen16_sub_rows=sub_opts[1], screen16_sublists=sub_opts[2],
                        screen16_prune=sub_opts[3], screen16_tighten=sub_opts[4], screen16c_qb=sub_opts[5])
            np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{int(time.time())}.npz"), centroids=a["centroids"],
                     list_len=a["list_len"], rows=a["rows"], tids=a["tids"], q=q, half=half, k=k, nprobe=nprobe, cap=cap,
                     strategy=strategy, mode=mode, opt_names=np.array(list(opts)), opt_values=np.array(list(opts.values())),
                     rows_f16=h if half else np.zeros(0, np.uint16))
            raise
    if not half and rng.random() < 0.25:
        check(lib.ndbhip_set_scan_mode(3))
        sharded_case(rng, ix, img, q, strategy, nprobe, k, cap, et, ed, ec)
        kind = kind + "/sharded"
    check(lib.ndbhip_set_scan_mode(0))
    ix.close()
    return kind


def main():
    from neurondb_amd import IvfIndex, _lib
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    _lib.ensure_init(0)
    _lib.use_torch_stream()
    rng = np.random.default_rng(seed)
    t0, n, kinds = time.time(), 0, {}
    while time.time() - t0 < secs:
        kd = one_case(rng, _lib.lib(), IvfIndex, _lib.check)
        kinds[kd] = kinds.get(kd, 0) + 1
        n += 1
    print(f"fuzz_scan: {n} random cases x 5 scan modes identical to the oracle (seed {seed}): {kinds}")


if __name__ == "__main__":
    main()
