#!/usr/bin/env python3
"""bench.py — kNN queries/sec on the IVFFlat list-scan hot path (BASELINE.json configs[1]:
1M x 768 fp32, lists=1024, probes=32, k=10, L2) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (centroid scan -> probe select -> list scan -> top-k)
over one batch of --batch synthetic queries, inputs resident in HBM.  N > 1 (--shard): a table
that fits one device several times over is REPLICATED and every rank answers its own batches —
queries are the independent units of this path, a step costs 1.8 ms and most of it is per-query
work that list sharding cannot divide — so `value` = N batches per step, weak
scaling, no data-path collective; the SHARDED path (lists cut over the ranks, every batch merged
over the library's RCCL communicator: what a table too large for one device needs) then runs
after the timed region and is checked against the oracle (`sharded_leg`).  --shard slices|lists
makes the sharded path the timed one (strong scaling: same index and query stream as N = 1).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# steps in flight on streams of their own (--inflight) must not share a hardware queue, where they would run one after
# the other: the HIP runtime maps a process's streams onto 4 queues unless told otherwise (read when it initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
# fp32 vector peak is 157.3 TFLOP/s with FMA (64 FLOP/clk/SIMD); the reference's recipe needs separately
# rounded subtract, multiply and add (no FMA), i.e. 32 FLOP/clk/SIMD = 78.6 TFLOP/s
UNFUSED_FP32_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD x 4 x 256 CUs x 2.4 GHz
FP16_MFMA_PEAK_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_f16, dense (MI355X_MICROARCH.md: ~2.5 PF; AMD's 5 PF is 2:1 sparse)
PCIE_PEAK_GBPS = 64.0                # host link: PCIe 5.0 x16 per direction (raw); measured pageable hipMemcpy: 56 GB/s


LINE_LIMIT = 6000         # bytes of the stdout line (the driver keeps an 8 KB tail of stdout: a longer line is UNMEASURED)


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return default
        d = d[k]
    return d


def _short_kernel(name):
    """`k_s16c_wsweep (centred one-plane sweep ...)` -> `k_s16c_wsweep`; at most 80 characters"""
    if not name:
        return None
    return str(name).split(" (")[0].split(";")[0][:80]


def _roof(r, extra=()):
    """the contract's roofline object, numbers only (+ the kernel's NAME): what the judge recomputes from"""
    if not isinstance(r, dict):
        return None
    out = {"bound": r.get("bound"), "kernel": _short_kernel(r.get("kernel")), "achieved": r.get("achieved"), "peak": r.get("peak"),
           "unit": r.get("unit"), "frac": r.get("frac"), "traffic": r.get("traffic"), "avg_launch_ms": r.get("avg_launch_ms"),
           "launches": r.get("launches")}
    for k in extra:
        v = _get(r, *k.split("."))
        if v is not None:
            out[k.replace(".", "_")] = v
    return out


def _leg(d, *keys):
    """one table's leg: queries/s, the sweep's fraction of its roof, recall and oracle mismatches"""
    if not isinstance(d, dict):
        return None
    if "error" in d or "skipped" in d:
        return {k: str(d[k])[:160] for k in ("error", "skipped") if k in d}
    out = {"queries_per_s": d.get("queries_per_s"), "ms_per_step": d.get("ms_per_step", d.get("ms_per_batch")),
           "bound": _get(d, "roofline", "bound"), "frac": _get(d, "roofline", "frac"),
           "kernel": _short_kernel(_get(d, "roofline", "kernel")), "recall_at_10": d.get("recall_at_10"),
           "oracle_mismatches": _get(d, "oracle_parity", "mismatches")}
    for k in keys:
        v = _get(d, *k.split("."))
        if v is not None:
            out[k.replace(".", "_")] = v
    return {k: v for k, v in out.items() if v is not None}


def driver_line(full, detail_path="bench_detail.json"):
    """The ONE line bench.py prints: the contract's fields, the headline's roofline and cpu_baseline as numbers, and one
    small object per leg.  Everything else bench.py measures (every note, every sub-leg, the library's counters) is in
    `full`, which main() writes to bench_detail.json beside this file.  Pure: no device, no files — tests/test_bench_line.py
    builds it from a committed detail file and checks size (< LINE_LIMIT) and strictness (no NaN / Infinity tokens)."""
    cfg = full.get("config") or {}
    cb = full.get("cpu_baseline")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype")}
    line["data"] = str(full.get("data", "synthetic"))[:120]
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], "data": str(cfg.get("data", ""))[:80],
                      "steps_in_flight": cfg.get("steps_in_flight_n", 1), "sharding": str(cfg.get("sharding", "none"))[:160],
                      "collectives": (None if not cfg.get("collectives") else
                                      {k: cfg["collectives"].get(k) for k in ("rccl_ranks", "per_step", "bytes_per_step_per_rank")}),
                      "shard": cfg.get("shard")}
    line["recall_at_10"] = full.get("recall_at_10")
    line["step_latency_ms"] = full.get("step_latency_ms")
    line["roofline"] = _roof(full.get("roofline"), ("hbm.frac", "mfma.frac", "hbm.step_frac", "hbm.bytes_per_launch", "mfma.flops_per_launch",
                                                   "hbm.traffic_over_bytes", "alone.frac", "alone.avg_launch_ms", "rows_emitted_per_query",
                                                   "rows_rescored_per_query"))
    line["cpu_baseline"] = None if not cb else {
        "value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
        "sample": str(cb.get("sample", ""))[:140], "gpu_mismatches_on_sample": _get(cb, "gpu_parity_on_sample", "mismatches"),
        "sample_queries": _get(cb, "gpu_parity_on_sample", "queries")}
    line["serial"] = None if not full.get("serial") else {k: full["serial"].get(k) for k in ("queries_per_s", "ms_per_step", "lanes_identical_to_serial")}
    # the other table of the same shape, at the top level with its own roofline
    for k in ("value_clustered", "ms_per_step_clustered", "recall_at_10_clustered", "value_iid", "recall_at_10_iid"):
        if full.get(k) is not None:
            line[k] = full[k]
    if full.get("roofline_clustered"):
        line["roofline_clustered"] = _roof(full["roofline_clustered"], ("hbm.step_frac", "alone.frac"))
    if full.get("iid_gauss"):
        line["iid_gauss"] = _leg(full["iid_gauss"])
    b = full.get("build")
    line["build_vectors_per_s"] = full.get("build_vectors_per_s")
    if b:
        line["build"] = {"vectors_per_s": b.get("vectors_per_s"), "searchable_vectors_per_s": b.get("searchable_vectors_per_s"),
                         "from_host_vectors_per_s": b.get("from_host_vectors_per_s"), "mfma_frac": _get(b, "roofline", "frac"),
                         "lists_identical_to_exact_assignment": b.get("lists_identical_to_exact_assignment"),
                         "cpu_vectors_per_s": _get(b, "cpu_baseline", "value"), "cpu_cores": _get(b, "cpu_baseline", "cores")}
    for k in ("replicated",):
        if full.get(k):
            line[k] = {kk: full[k].get(kk) for kk in ("queries_per_s", "ms_per_step")}
    if full.get("dist_parity_on_sample"):
        line["dist_parity_on_sample"] = full["dist_parity_on_sample"]
    if full.get("sharded_leg"):
        line["sharded_leg"] = _leg(full["sharded_leg"], "dist_parity_on_sample") if "error" not in full["sharded_leg"] else \
            {"error": str(full["sharded_leg"]["error"])[:160]}
    for k in ("balanced_index", "c4", "c5"):
        if full.get(k):
            line[k] = _leg(full[k], "serial.queries_per_s", "exact_scan_parity.mismatches")
    if full.get("sigma_sweep"):
        line["sigma_sweep"] = {name.replace("sigma_", "s").replace("anisotropic_", "aniso_"):
                               ({"error": str(v["error"])[:80]} if "error" in v else
                                {"queries_per_s": v.get("queries_per_s"), "bound": _get(v, "roofline", "bound"), "frac": _get(v, "roofline", "frac"),
                                 "kernel": _short_kernel(_get(v, "roofline", "kernel")), "recall_at_10": v.get("recall_at_10"),
                                 "oracle_mismatches": _get(v, "oracle_parity", "mismatches")})
                               for name, v in full["sigma_sweep"].items()}
    h = full.get("hnsw")
    if h:
        if "error" in h:
            line["hnsw"] = {"error": str(h["error"])[:160]}
        else:
            line["hnsw"] = _leg(h, "build_vectors_per_s", "strategy", "evaluations_per_query", "one_batch_at_a_time.queries_per_s",
                                "intended.oracle_parity.mismatches", "intended.cpu_baseline.value", "intended.iid_gauss_unit.recall_at_10_by_ef.64")
            line["hnsw"]["ref_compat"] = {"queries_per_s": _get(h, "ref_compat", "queries_per_s"), "recall_at_10": _get(h, "ref_compat", "recall_at_10"),
                                          "oracle_mismatches": _get(h, "ref_compat", "oracle_parity", "mismatches")}
    if full.get("note"):
        line["note"] = str(full["note"])[:160]
    line["detail"] = detail_path
    s = json.dumps(line, allow_nan=False)
    if len(s) > LINE_LIMIT:                       # (never print what the driver cannot read: drop legs, keep the contract)
        for k in ("sigma_sweep", "balanced_index", "iid_gauss", "build", "c5", "c4", "hnsw", "sharded_leg"):
            line.pop(k, None)
            line["truncated"] = True
            if len(json.dumps(line, allow_nan=False)) <= LINE_LIMIT:
                break
    return line


def _finite(o):
    """NaN / +-Infinity are not JSON: they become null in what bench.py writes"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating,)):
        return _finite(float(o))
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def emit(full, json_fd):
    """bench_detail.json (everything) beside bench.py and under gpurun_out/ when that exists; the short line on stdout"""
    full = _finite(full)
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, allow_nan=False)
                f.write("\n")
        except OSError as e:                      # (a read-only checkout must not cost the line)
            sys.stderr.write(f"[bench] could not write {p}: {e}\n")
    line = driver_line(full)
    os.write(json_fd, (json.dumps(line, allow_nan=False) + "\n").encode())
    return line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nvec", type=int, default=None,
                    help="rows of the table.  Default: 1 000 000 (BASELINE.json configs[1], the workload `metric` is quoted on) at "
                         "--gpus 1; at --gpus N > 1, 10 000 000 with --lists 4096 (configs[3]: the table north_star shards over the "
                         "ranks, lists cut into slices, every batch merged over RCCL inside the timed region)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--lists", type=int, default=None, help="default 1024 (N = 1) / 4096 (N > 1), see --nvec")
    ap.add_argument("--probes", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="queries per step")
    ap.add_argument("--inflight", type=int, default=None,
                    help="default: 3.  N = 1, unsharded: steps in flight at once — that many mirrors of the index, each driven by a host thread "
                         "with a stream of its own (ndbhip_set_thread_stream): a step's per-query chains (selection, seeds, pair "
                         "tables, finalize: waves waiting for memory) run under another step's sweep; the sweeps themselves queue "
                         "up.  1 = one step after the other (reported either way as `serial`)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--build-from-host", type=int, default=1, help="also time ndbhip_ivf_build from host memory (0 = skip)")
    ap.add_argument("--recall-queries", type=int, default=200)
    ap.add_argument("--data", choices=["clustered", "gauss"], default=None,
                    help="gauss: i.i.d. N(0,1) rows and queries — BASELINE.md section 2's data, the default at --gpus 1 (`value`); "
                         "clustered: mixture of --components Gaussians (sigma --sigma), SURVEY 8d's optional variant — "
                         "`value_clustered` of the default line, and the default table of the sharded N > 1 run")
    ap.add_argument("--components", type=int, default=None, help="default: one per list")
    ap.add_argument("--sigma", type=float, default=0.1)
    ap.add_argument("--rows", choices=["f32", "f16"], default="f32",
                    help="f16: search a halfvec twin of the index (rows narrowed round-to-nearest-even on the device)")
    ap.add_argument("--f16-encoder", choices=["rne", "reference"], default="rne",
                    help="--rows f16: narrow with round-to-nearest-even, or with the reference's float4_to_fp16 "
                         "(truncating, subnormals flushed)")
    ap.add_argument("--strategy", choices=["l2", "cosine", "ip"], default="l2")
    ap.add_argument("--shard", choices=["auto", "replicas", "slices", "lists"], default="auto",
                    help="N > 1: replicas = every rank holds the whole index and answers its own batches (weak scaling, no "
                         "data-path collective); slices / lists = the index is sharded (heavy lists cut into per-rank slices, "
                         "or lists kept whole) and every batch is merged over RCCL (strong scaling); auto = replicas when "
                         "four copies of the table fit half the device memory, else slices.  With replicas the sharded "
                         "path is still run and checked after the timed region (`sharded_leg`)")
    ap.add_argument("--gauss-steps", type=int, default=3,
                    help="N=1, default workload: also run this many steps on i.i.d. N(0,1) data (0 = skip)")
    ap.add_argument("--c5-nvec", type=int, default=10_000_000,
                    help="N=1, default workload: also run BASELINE.md's C5 (halfvec x 1536, inner product, lists 4096, batches "
                         "of 256) on this many rows on the one GPU (0 = skip; skipped with a note when the device has less "
                         "free memory than 4.5 x the table: the leg holds the fp32 rows, their fp16 twin and the planes at once)")
    ap.add_argument("--c4-nvec", type=int, default=10_000_000,
                    help="N=1, default workload: also run BASELINE.md's C4 shape (x 768 fp32, lists 4096, L2, 4096 queries a step) "
                         "on this many rows on the one GPU (0 = skip)")
    ap.add_argument("--sigma-sweep", type=int, default=1,
                    help="N=1, default workload: the headline shape at sigma 0.1 ... 1.0 and one anisotropic table (0 = skip)")
    ap.add_argument("--hnsw-nvec", type=int, default=1_000_000,
                    help="also measure HNSW build + search (BASELINE config C3) on this many rows at N=1 (0 = skip)")
    ap.add_argument("--dist-parity-queries", type=int, default=128,
                    help="N>1: merged results of this many queries are checked against the CPU oracle on rank 0")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="ndbhip_set_option(NAME, VALUE) before the run (A/B of library switches), repeatable")
    ap.add_argument("--dist-impl", choices=["c", "torch"], default="c",
                    help="N > 1: the exchange inside the C library (ndbhip_ivf_search_sharded, RCCL from C) or the "
                         "same steps as torch.distributed calls (neurondb_amd.dist.sharded_search)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded path (process group, partial search, all-gather, merge) even at N=1")
    return ap.parse_args()


def pack_tids(rows: torch.Tensor) -> torch.Tensor:
    """row number -> device TID format (uint64 image of ItemPointerData): block = row // 64, offset = row % 64 + 1."""
    blk = rows // 64
    bi_hi = (blk >> 16) & 0xFFFF
    bi_lo = blk & 0xFFFF
    pos = rows % 64 + 1
    return (bi_hi | (bi_lo << 16) | (pos << 32)).to(torch.int64)


def unpack_tids(t: torch.Tensor) -> torch.Tensor:
    bi_hi = t & 0xFFFF
    bi_lo = (t >> 16) & 0xFFFF
    pos = (t >> 32) & 0xFFFF
    return ((bi_hi << 16) | bi_lo) * 64 + pos - 1


def make_data(n, dim, kind, components, sigma, seed, center_seed, dev):
    """Synthetic vectors from the in-repo counter-based generator (csrc/ndbhip_gen.h): element
    (row, d) is a pure function of the seeds, identical on every rank, and ndbhip_gen_rows_host regenerates the
    same bits without a device.  clustered: x = center[comp(row)] + sigma * N(0,1) with centers ~ N(0,1) shared by
    base and queries (SURVEY 8d 'clustered variant'); gauss: i.i.d. N(0,1)."""
    import ctypes as C
    from neurondb_amd._lib import check, lib
    x = torch.empty((n, dim), dtype=torch.float32, device=dev)
    check(lib().ndbhip_gen_rows_device(1 if kind == "clustered" else 0, seed, center_seed, 0, n, dim, components,
                                       float(sigma), C.c_void_p(x.data_ptr())))
    check(lib().ndbhip_synchronize())
    return x


def main():
    args = parse()
    # Only the JSON line may reach stdout: RCCL / HIP runtime banners written by C libraries to fd 1
    # (some are flushed at exit, after Python's own output) are sent to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.data is None:
        args.data = "clustered" if (world > 1 or args.force_dist) else "gauss"
    lanes_legs = 3 if args.inflight is None else max(1, args.inflight)       # the clustered legs' steps in flight
    args.lanes_legs = lanes_legs
    if args.inflight is None:
        # (round 6: the i.i.d. table too — its dense sweep fills the device, but the 0.8 ms of chain kernels around it run under
        # the next step's sweep: 6.10 -> 5.76 ms a step with three in flight, measured on one box)
        args.inflight = 3
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    comm_error = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd.dist import ShardedSearchBuffers, partition_lists, partition_slices, sharded_search
    from neurondb_amd._lib import check, lib
    _lib.ensure_init(local_rank)
    # one stream for torch (data, RCCL collectives) and the library's kernels: the device-pointer calls are
    # asynchronous, and their inputs / outputs are produced / consumed by torch
    stream = _lib.use_torch_stream()
    for o in args.opt:
        name, value = o.split("=")
        check(lib().ndbhip_set_option(name.encode(), int(value)))
    if use_dist and args.dist_impl == "c":
        # the exchange runs inside the C library (ndbhip_ivf_search_sharded over its own RCCL communicator);
        # torch.distributed only carries the 128-byte unique id and the timing barrier
        from neurondb_amd.dist import init_library_comm
        try:
            init_library_comm(device=dev)
        except Exception as e:              # (init_library_comm agrees on the outcome: every rank raises or none does)
            comm_error = f"{type(e).__name__}: {e}"
            args.dist_impl = "torch"        # the same exchange through torch.distributed (neurondb_amd/dist.py)

    # the workload: N = 1 -> C2 (what `metric` is quoted on); N > 1 -> C4, SHARDED (north_star: "IVFFlat list scans shard
    # naturally across the 8 GPUs of one node with RCCL top-k merge"): the collective is inside the timed region, N = 1 of the
    # same table is the `c4` key of the N = 1 line, and the replicated number (no collective) is a sub-key
    c4_default = world > 1 and args.nvec is None and args.lists is None and not args.force_dist
    if args.nvec is None:
        args.nvec = 10_000_000 if world > 1 else 1_000_000
    if args.lists is None:
        args.lists = 4096 if c4_default else 1024
    if args.components is None:
        args.components = args.lists
    n, dim, nlists, nprobe, k, nq = args.nvec, args.dim, args.lists, args.probes, args.k, args.batch
    shard_mode = args.shard
    if shard_mode == "auto":
        fits = 4.0 * n * dim * 4 < 0.5 * torch.cuda.get_device_properties(dev).total_memory
        shard_mode = "slices" if (c4_default or not fits) else "replicas"
    sharded = use_dist and shard_mode in ("slices", "lists")       # the timed step is the RCCL-merged sharded search
    replicas = use_dist and not sharded                              # the timed step is each rank's own batch on a full copy

    # ---------------- synthetic data (identical on every rank) ----------------
    base = make_data(n, dim, args.data, args.components, args.sigma, 0x5EED0001, 0x5EEDC0DE, dev)
    nq_total = nq * (args.steps + args.warmup)
    # (replicas: every rank answers its own queries; rank 0's are the N = 1 run's)
    queries = make_data(max(nq_total, args.recall_queries), dim, args.data, args.components, args.sigma,
                        0x5EED0002 + (0x100000 * rank if replicas else 0), 0x5EEDC0DE, dev)

    # ---------------- index build: the product's build path, timed ----------------
    # sample first min(10000, 100*lists) rows + k-means (reference rule) + assign all rows + pack lists
    tids_all = pack_tids(torch.arange(n, device=dev))
    # untimed warm-up, like the W warm-up steps of the search: one build of the same table.  The first build of a
    # process pays the code-object loads and, on a fresh box, the runtime's first multi-GB allocations (0.3 ... 60 ms
    # each for the same call on the same box); its wall time is reported next to the timed one as `first_build_seconds`
    warm = IvfIndex(dim, nlists, device=local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    warm.build_device(base, tids_all, 50)
    check(lib().ndbhip_synchronize())
    t_first_build = time.perf_counter() - t0
    warm.close()
    ix_full = IvfIndex(dim, nlists, device=local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kmeans_iters = ix_full.build_device(base, tids_all, 50)
    check(lib().ndbhip_synchronize())
    t_build = time.perf_counter() - t0
    build_vps = n / t_build
    # ... and what makes the index searchable at full speed: sublists, fp16 planes, norms, radii (ndbhip_ivf_prepare;
    # otherwise the first batched scan pays for it)
    t0 = time.perf_counter()
    if args.rows == "f32":
        ix_full.prepare({"l2": 1, "cosine": 2, "ip": 3}[args.strategy])
    check(lib().ndbhip_synchronize())
    t_prepare = time.perf_counter() - t0
    cent_h, list_len, _, _ = ix_full.export(rows=False)
    # ... and the same build from where a CREATE INDEX finds its table: host memory (pageable, like a backend's
    # palloc'd tuples).  ndbhip_ivf_build uploads through its pinned lanes while the k-means runs; the rate includes
    # PCIe, the build and the prepare — what the reference's ivfbuild is timed as, whole (rank 0, one GPU)
    from_host = None
    if rank == 0 and world == 1 and args.build_from_host:
        host_rows = base.cpu().numpy()
        host_tids = tids_all.cpu().numpy()
        hx = IvfIndex(dim, nlists, device=local_rank)
        hx.build(host_rows, host_tids, 50)       # untimed warm-up of the same call, like the device build's above
        hx.close()
        hx = IvfIndex(dim, nlists, device=local_rank)
        t0 = time.perf_counter()
        hx.build(host_rows, host_tids, 50)
        t_hb = time.perf_counter() - t0
        t0 = time.perf_counter()
        hx.prepare({"l2": 1, "cosine": 2, "ip": 3}[args.strategy])
        check(lib().ndbhip_synchronize())
        t_hp = time.perf_counter() - t0
        _, hl, _, _ = hx.export(rows=False)
        from_host = {"vectors_per_s": round(n / t_hb, 1), "searchable_vectors_per_s": round(n / (t_hb + t_hp), 1),
                     "build_seconds": round(t_hb, 4), "prepare_seconds": round(t_hp, 4), "host_bytes": int(n) * dim * 4,
                     "same_lists_as_device_build": bool(np.array_equal(hl, list_len)),
                     "roofline": {"bound": "pcie", "achieved": round(n * dim * 4 / t_hb / 1e9, 2), "peak": PCIE_PEAK_GBPS,
                                  "unit": "GB/s", "frac": round(n * dim * 4 / t_hb / 1e9 / PCIE_PEAK_GBPS, 4),
                                  "note": "table bytes / build_seconds against the host link (PCIe 5.0 x16, 64 GB/s raw; "
                                          "a bare hipMemcpy of the same pageable table reaches 56 GB/s on this box: "
                                          "profiles/r03_host_build_timeline.txt): the rows arrive in heap order while "
                                          "the k-means and the assignment of the slabs already there run, so the build "
                                          "adds only its last slab and the list packing to the transfer"},
                     "note": "ndbhip_ivf_build(host rows, pageable) then ndbhip_ivf_prepare: wall time from the call to "
                             "an index that answers (vectors_per_s) and one that answers at full speed "
                             "(searchable_vectors_per_s)"}
        hx.close()
        del host_rows, host_tids
    build = None
    if rank == 0:
        # SURVEY 8d "IVF build roofline": the assignment of all N rows is 2 N lists dim flops (a dense contraction;
        # k-means on the 10 000-row sample adds iterations x 10 000 rows of the same)
        flops = 2.0 * (n + kmeans_iters * min(10000, 100 * nlists, n)) * nlists * dim
        # the same rows through both assignment paths of the library (outside the timed region): the screened
        # matrix-core path the build used and the exact vector-ALU path must give the same list for every row
        cent_d = torch.from_numpy(cent_h).to(dev)
        asg = {}
        for opt in (1, 0):
            check(lib().ndbhip_set_option(b"build_screen16", opt))
            out = torch.full((n,), -1, dtype=torch.int32, device=dev)
            check(lib().ndbhip_ivf_assign_device(cent_d.data_ptr(), nlists, dim, base.data_ptr(), n, out.data_ptr()))
            check(lib().ndbhip_synchronize())
            asg[opt] = out
        check(lib().ndbhip_set_option(b"build_screen16", 1))
        for kv in args.opt:                              # (an explicit --opt build_screen16=... stays in force)
            name, _, val = kv.partition("=")
            if name == "build_screen16":
                check(lib().ndbhip_set_option(b"build_screen16", int(val)))
        same = bool(torch.equal(asg[0], asg[1]))
        counts_ok = bool(np.array_equal(torch.bincount(asg[0].long(), minlength=nlists).cpu().numpy(), list_len))
        del asg, cent_d
        build = {"vectors_per_s": round(build_vps, 1), "seconds": round(t_build, 4), "kmeans_iterations": int(kmeans_iters),
                 "first_build_seconds": round(t_first_build, 4),
                 "prepare_seconds": round(t_prepare, 4),
                 "searchable_vectors_per_s": round(n / (t_build + t_prepare), 1),
                 "from_host": from_host,
                 "from_host_vectors_per_s": from_host["vectors_per_s"] if from_host else None,
                 "lists_identical_to_exact_assignment": same and counts_ok,
                 "roofline": (lambda exe: {
                     "bound": "mfma", "achieved": round(exe / t_build / 1e12, 2), "peak": FP16_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(exe / t_build / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4),
                     "executed_flops": int(exe),
                     "algorithmic": {"flops": int(flops), "achieved": round(flops / t_build / 1e12, 2),
                                     "frac": round(flops / t_build / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4)},
                     "note": "whole build (sample k-means, assignment of every row, list packing) over the flops it EXECUTES, "
                             "against the dense fp16 matrix-core peak: the assignment runs as two sweeps of k_s16_sweep "
                             "(row minimum, then the centroids within the error bound of it), each 3 fp16 products per "
                             "multiply (split operands) = 12 N lists dim, plus the k-means iterations; `algorithmic` = "
                             "SURVEY 8d's 2 N lists dim.  The reference's fp32 arithmetic decides only among the "
                             "centroids the bound leaves (about 1 % of the rows have more than one)"})(
                     6.0 * 2.0 * n * nlists * dim + flops - 2.0 * n * nlists * dim),
                 "cpu_baseline": (build_cpu_baseline(args, base, cent_h, kmeans_iters) if args.cpu_seconds > 0 and world == 1
                                  else None)}
    del base
    full_image = None
    if use_dist and rank == 0 and args.dist_parity_queries > 0 and args.rows == "f32" and args.strategy == "l2":
        full_image = ix_full.export(rows=True)          # host copy of the unsharded index for the parity sample

    if args.rows == "f16":
        ix16 = ix_full.to_f16(reference_encoder=(args.f16_encoder == "reference"))
        ix_full.close()
        ix_full = ix16
    strategy = {"l2": 1, "cosine": 2, "ip": 3}[args.strategy]

    # ---------------- shard lists over ranks ----------------
    # Balanced by WORK: a list costs len x (queries probing it).  The probe counts come from a calibration batch
    # drawn like the queries but with its own seed (a deployment uses recent traffic); every rank computes the
    # same partition from the same inputs.  Lists too heavy for one rank are cut into slices.
    def make_shard(src, mode):
        """this rank's shard of `src` (a full mirror) and a description of the partition"""
        qcal = make_data(nq, dim, args.data, args.components, args.sigma, 0x5EED0007, 0x5EEDC0DE, dev)
        pcal = torch.zeros((nq, nprobe), dtype=torch.int32, device=dev)
        src.select_clusters_device(qcal, pcal, nprobe)
        check(lib().ndbhip_synchronize())
        pc = pcal.cpu().numpy()
        cnt = np.bincount(pc[pc >= 0].ravel(), minlength=len(list_len))[:len(list_len)]
        del qcal, pcal
        if mode == "slices":
            slo, sln, stl = partition_slices(list_len, world, cnt)
            sh = src.shard_slices(slo[rank], sln[rank], stl[rank])
            work = (sln * (cnt[None] + 1.0)).sum(1)
            info = {"by": "slices", "lists_cut": int(((sln > 0).sum(0) > 1).sum()),
                    "max_work_share": round(float(work.max() / work.sum()), 4)}
        else:
            owner = partition_lists(list_len, world, cnt)
            sh = src.shard((owner == rank).astype(np.uint8))
            work = np.bincount(owner, weights=np.asarray(list_len) * (cnt + 1.0), minlength=world)
            info = {"by": "lists", "max_work_share": round(float(work.max() / work.sum()), 4)}
        return sh, info

    shard_info = None
    replicated = None
    if sharded:
        ix, shard_info = make_shard(ix_full, shard_mode)
        if world > 1:
            # the sub-key: the same table REPLICATED (every rank the whole index, its own batches, no data-path collective) —
            # what the ranks can do when the table fits each of them; the timed step below is the sharded one
            rq_ = make_data(nq * (args.warmup + args.steps), dim, args.data, args.components, args.sigma,
                            0x5EED0002 + 0x100000 * rank, 0x5EEDC0DE, dev)
            rt_, rd_, rc_ = (torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
                             torch.zeros(nq, dtype=torch.int32, device=dev))
            for w in range(args.warmup):
                ix_full.search_device(rq_[w * nq:(w + 1) * nq], rt_, rd_, rc_, strategy, nprobe, k, 0)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s_ in range(args.steps):
                i_ = args.warmup + s_
                ix_full.search_device(rq_[i_ * nq:(i_ + 1) * nq], rt_, rd_, rc_, strategy, nprobe, k, 0)
            dist.barrier()
            torch.cuda.synchronize()
            tt_ = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            replicated = {"queries_per_s": round(world * nq * args.steps / float(tt_.item()), 1),
                          "ms_per_step": round(float(tt_.item()) / args.steps * 1e3, 3),
                          "note": f"every rank holds the whole table and answers its own {nq}-query batches: {world} batches a step, "
                                  f"no data-path collective (weak scaling); `value` is the SHARDED step"}
            del rq_, rt_, rd_, rc_
        ix_full.close()
        ix_full = None
        torch.cuda.empty_cache()
    else:
        ix = ix_full

    # ---------------- one step ----------------
    buf = ShardedSearchBuffers(nq, k, world, dev, nprobe=nprobe if use_dist else 0)
    out_t, out_d, out_c = buf.out_tids, buf.out_dist, buf.out_count

    def step(qs):
        if not sharded:
            ix.search_device(qs, out_t, out_d, out_c, strategy, nprobe, k, 0)
        elif args.dist_impl == "c":
            ix.search_sharded_device(qs, out_t, out_d, out_c, strategy, nprobe, k, 0)
        else:
            sharded_search(ix, qs, buf, strategy, nprobe, k, 0, rank=rank)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    inflight = max(1, args.inflight) if not use_dist else 1
    mirrors, lanes = [ix], []
    if inflight > 1:
        # the other steps in flight search SHARES of the mirror (ndbhip_ivf_share: the same rows, planes and tables, scratch
        # of their own); a share builds nothing, so the source runs one batch first (matrices, samples, constants)
        import threading
        for _ in range(2):              # (the second batch knows the first one's pairs per bucket: tile size, the rows' sample)
            step(queries[:nq])
        torch.cuda.synchronize()
        for _ in range(inflight - 1):
            mirrors.append(ix.share())
        for w in range(inflight):
            lanes.append({"stream": torch.cuda.Stream(), "t": torch.zeros_like(out_t), "d": torch.zeros_like(out_d),
                          "c": torch.zeros_like(out_c)})
        torch.cuda.synchronize()

    def run_steps(first, count):
        """steps first .. first + count - 1, `inflight` at a time: every lane takes the next step that nobody has taken yet
        (a lane that falls behind does not leave its share of the steps for the end)"""
        if inflight == 1:
            for s in range(first, first + count):
                step(queries[s * nq:(s + 1) * nq])
            return
        err = []
        nxt, nxt_lock = [first], threading.Lock()

        def lane(w):
            try:
                ln = lanes[w]
                check(lib().ndbhip_set_thread_stream(C.c_void_p(ln["stream"].cuda_stream)))
                while True:
                    with nxt_lock:
                        s = nxt[0]
                        nxt[0] += 1
                    if s >= first + count:
                        break
                    mirrors[w].search_device(queries[s * nq:(s + 1) * nq], ln["t"], ln["d"], ln["c"], strategy, nprobe, k, 0)
                    ln["last"] = s
                check(lib().ndbhip_synchronize())
            except Exception as e:              # (surfaced by the caller: a lane must not die silently)
                err.append(e)
            finally:
                lib().ndbhip_set_thread_stream(None)
        th = [threading.Thread(target=lane, args=(w,)) for w in range(inflight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]

    import ctypes as C
    run_steps(0, args.warmup)
    barrier()
    check(lib().ndbhip_stats_reset())
    check(lib().ndbhip_profile(1))
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    check(lib().ndbhip_profile(0))
    st = _lib.stats()
    serial, st_serial = None, None
    if inflight > 1:
        # the same steps one after the other on the first mirror: what rounds 1-4 timed; its results are the reference the
        # lanes' last steps are compared with
        save = inflight
        lane_out = [(lanes[w]["t"].clone(), lanes[w]["d"].clone(), lanes[w]["c"].clone()) for w in range(inflight)]
        lane_last = [lanes[w].get("last") for w in range(inflight)]
        inflight = 1
        run_steps(0, args.warmup)
        barrier()
        check(lib().ndbhip_stats_reset())
        check(lib().ndbhip_profile(1))
        t0 = time.perf_counter()
        run_steps(args.warmup, args.steps)
        barrier()
        el1 = time.perf_counter() - t0
        check(lib().ndbhip_profile(0))
        st_serial = _lib.stats()
        inflight = save
        # every lane's last step of the timed region, rerun one at a time and compared
        same = True
        for w in range(inflight):
            last = lane_last[w]
            if last is None or last < args.warmup:
                continue
            step(queries[last * nq:(last + 1) * nq])
            torch.cuda.synchronize()
            same = same and bool(torch.equal(out_t, lane_out[w][0]) and torch.equal(out_d.view(torch.int32), lane_out[w][1].view(torch.int32))
                                 and torch.equal(out_c, lane_out[w][2]))
        serial = {"queries_per_s": round(nq * args.steps / el1, 1), "ms_per_step": round(el1 / args.steps * 1e3, 4),
                  "lanes_identical_to_serial": same,
                  "note": "the same steps one after the other on one stream and one mirror (what rounds 1-4 reported as `value`)"}
        for m in mirrors[1:]:
            m.close()
        mirrors, lanes, lane_out = [ix], [], None
        torch.cuda.empty_cache()
    if os.environ.get("NDB_PHASES"):
        # a profiling build of the library (make EXTRA=-DNDB_PHASES): block 0's clock stamps of the last step's kernels
        import ctypes as _C
        ph = (_C.c_ulonglong * 64)()
        check(lib().ndbhip_debug_phases(ph))
        ph = list(ph)
        sys.stderr.write("[phases] cent_select us: " + " ".join(f"{(ph[i + 1] - ph[i]) / 100:.1f}" for i in range(0, 7)) +
                         "   finalize us: " + " ".join(f"{(ph[i + 1] - ph[i]) / 100:.1f}" for i in range(16, 21)) + "\n")
    if os.environ.get("NDB_TRACE"):
        # a profiling build (make EXTRA=-DNDB_PHASES): the register-streaming sweep's per-wave trace of the last step
        import ctypes as _C
        nw = 24576
        tr = (_C.c_ulonglong * nw)()
        check(lib().ndbhip_debug_trace(tr, nw))
        tr = np.array(list(tr), dtype=np.int64).reshape(-1, 12)
        tr = tr[tr[:, 1] > 0]
        if len(tr):
            t0w, t1w = tr[:, 0].min(), tr[:, 1].max()
            span = (t1w - t0w) / 100.0
            ends = (tr[:, 1] - t0w) / 100.0
            starts = (tr[:, 0] - t0w) / 100.0
            busy = (tr[:, 1] - tr[:, 0]) / 100.0
            sys.stderr.write(f"[trace] waves {len(tr)} span {span:.1f} us; start p50/p99/max {np.percentile(starts, 50):.1f}/"
                             f"{np.percentile(starts, 99):.1f}/{starts.max():.1f}; end min/p10/p50/p90/max {ends.min():.1f}/"
                             f"{np.percentile(ends, 10):.1f}/{np.percentile(ends, 50):.1f}/{np.percentile(ends, 90):.1f}/{ends.max():.1f}; "
                             f"busy mean {busy.mean():.1f}; wait share {tr[:, 3].sum() / max(1, (tr[:, 1] - tr[:, 0]).sum()):.3f}; "
                             f"items/wave min/mean/max {tr[:, 2].min()}/{tr[:, 2].mean():.2f}/{tr[:, 2].max()}; "
                             f"us per item {busy.sum() / max(1, tr[:, 2].sum()):.2f}; emission path: share {tr[:, 4].sum() / max(1, (tr[:, 1] - tr[:, 0]).sum()):.3f}, "
                             f"items taking it {tr[:, 5].sum() / max(1, tr[:, 2].sum()):.3f}, us each {tr[:, 4].sum() / 100.0 / max(1, tr[:, 5].sum()):.2f}; "
                             f"looking for work elsewhere: share {tr[:, 6].sum() / max(1, (tr[:, 1] - tr[:, 0]).sum()):.3f}; "
                             f"emission path us per taking item: members' constants {tr[:, 8].sum() / 100.0 / max(1, tr[:, 5].sum()):.2f}, "
                             f"per-element test {tr[:, 9].sum() / 100.0 / max(1, tr[:, 5].sum()):.2f}, counts + slots {tr[:, 10].sum() / 100.0 / max(1, tr[:, 5].sum()):.2f}, "
                             f"records {tr[:, 11].sum() / 100.0 / max(1, tr[:, 5].sum()):.2f}; a wave's longest item us p50/p90/p99/max "
                             f"{np.percentile(tr[:, 7], 50) / 100.0:.1f}/{np.percentile(tr[:, 7], 90) / 100.0:.1f}/{np.percentile(tr[:, 7], 99) / 100.0:.1f}/{tr[:, 7].max() / 100.0:.1f}\n")
            np.save(os.environ["NDB_TRACE"], tr)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    qps = nq * args.steps * (world if replicas else 1) / elapsed       # whole job: replicas answer `world` batches a step

    # ---------------- roofline of the dominant kernel ----------------
    grouped = (nq >= 5 and dim % 64 == 0)
    screened = st.get("rows_rescored", 0) > 0                 # the L2 scan ran as bound + exact second pass
    recipe = {"l2": "R_IVF_L2", "cosine": "R_IVF_COS", "ip": "R_IVF_IP"}[args.strategy]
    if screened:
        recipe = {"l2": "R_SCR_L2", "cosine": "R_SCR_COS", "ip": "R_SCR_IP"}[args.strategy]
    esz = 2 if args.rows == "f16" else 4
    kernel = (f"k_ivf_scan_grouped<{recipe}{', fp16 rows' if esz == 2 else ''}>" if grouped
              else f"k_ivf_scan{'_h' if esz == 2 else ''}<{recipe}>")
    opts = dict(o.split("=") for o in args.opt)          # the library switches this run set (ndbhip_set_option)
    scr_coop, scr_mfma = opts.get("scr_coop", "2"), opts.get("scr_mfma", "1")
    if screened and scr_coop != "0" and dim % 16 == 0:
        kernel = (f"k_ivf_bound_coop2<{recipe}>" if scr_coop == "2" else "k_ivf_bound_coop<R_SCR_L2>")
    mfma = screened and scr_coop == "2" and dim % 16 == 0 and scr_mfma != "0"
    if mfma:
        kernel = f"k_ivf_bound_mfma<{recipe}>"
    launches = max(1, st["scan_launches"])
    bytes_per_launch = st["bytes_scored"] / launches          # algorithmic: probed rows x dim x 4 B per query
    ms_per_launch = st["scan_kernel_ms"] / launches
    achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    # exact recipes: every scored (row element, query) pair costs one subtract, one multiply, one add, unfused
    # (inner product: multiply + add; cosine: the same plus the row's own norm chain, shared by the 16 queries
    # of a group — the query's norm is computed once per query).  Screened scans: one fused multiply-add per pair
    # element (the rows' norms are kept per row); the fused peak is twice the unfused one.
    per_elem = 2.0 if screened else {"l2": 3.0, "ip": 2.0, "cosine": 2.0 + 2.0 / 16.0}[args.strategy]
    valu_peak = 2.0 * UNFUSED_FP32_PEAK_TFLOPS if screened else UNFUSED_FP32_PEAK_TFLOPS
    flops_per_launch = per_elem * bytes_per_launch / esz
    valu_tflops = flops_per_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
    if screened:
        note = ("algorithmic bytes = rows scored x 3072 B per query.  Screened scan: a fused-multiply-add pass bounds every "
                "candidate's distance from below for up to 16 queries per staged row tile; the candidates that can still be "
                f"among the k nearest ({st['rows_rescored'] / max(1, nq * args.steps):.0f} per query) get the reference's "
                "sequential arithmetic in a second pass, so ids, ranks and float4 bits are the exact path's.  One block "
                "scores 128 rows against 4 query groups (64 queries) from one staged tile")
    elif grouped:
        note = ("algorithmic bytes = rows scored x 3072 B per query; the grouped kernel stages each row tile once "
                "for up to 16 queries, so HBM traffic is ~1/16 of that and the limiter is the fp32 vector ALU")
    else:
        note = "one pass over the probed rows per query"
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": pmc_traffic(args, world, kernel.split("<")[0])[0],
                "traffic_source": pmc_traffic(args, world, kernel.split("<")[0])[1], "bytes_per_launch": int(bytes_per_launch),
                "avg_launch_ms": round(ms_per_launch, 4), "launches": int(launches),
                "rows_rescored_per_query": round(st.get("rows_rescored", 0) / max(1, nq * args.steps), 1),
                "note": note,
                "valu": {"achieved": round(valu_tflops, 2), "peak": valu_peak, "unit": "TFLOP/s",
                         "frac": round(valu_tflops / valu_peak, 4),
                         "ops": "fused multiply-add" if screened else "unfused subtract / multiply / add"}}

    if mfma:
        # The bound pass runs on the matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fmaf chain, dense
        # fp32 MFMA peak 157.3 TFLOP/s = 64 FLOP/clk/SIMD at 2.4 GHz, MI355X_MICROARCH.md): that is the roof
        # that bounds it.  flops = 2 per (row element, query) pair actually scored; the algorithmic-bytes figure
        # of SURVEY 8d (no reuse across queries) and the measured HBM traffic stay alongside.
        roofline = {"bound": "mfma", "kernel": kernel, "achieved": round(valu_tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(valu_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": roofline["traffic"], "traffic_source": roofline["traffic_source"],
                    "flops_per_launch": int(flops_per_launch),
                    "bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(ms_per_launch, 4),
                    "launches": int(launches),
                    "rows_rescored_per_query": roofline["rows_rescored_per_query"],
                    "note": ("bound pass of the screened scan on fp32 MFMA: one block scores 128 rows against 64 queries "
                             "(4 groups) from one staged LDS tile; the MFMA's k-ordered fmaf chain is the chain the "
                             "vector-ALU bound kernel ran, so the lower bounds have the same bits and the "
                             f"{roofline['rows_rescored_per_query']:.0f} candidates per query that can still be among the "
                             "k nearest get the reference's sequential arithmetic in a second pass: ids, ranks and float4 "
                             "bits are the exact path's.  flops = 2 x dim per scored (row, query) pair"),
                    "hbm_algorithmic": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": round(achieved / HBM_PEAK_GBPS, 4),
                                        "note": "rows scored x row bytes per query (SURVEY 8d, no reuse across queries); "
                                                "the kernel reuses a staged tile for 64 queries, so the real HBM "
                                                "traffic is `traffic`"}}

    s16 = st.get("screen16_batches", 0) > 0 and st.get("screen16_fallbacks", 0) == 0
    # (plane_bytes is counted by the centred one-plane sweep only: L2, and inner product / cosine on the same planes)
    centred = s16 and st.get("plane_bytes", 0) > 0 and \
        dict(o.split("=") for o in args.opt).get("screen16_centered", "1") != "0"
    if centred:
        roofline = sweep_roofline(args, st, nq, args.steps, elapsed / args.steps * 1e3, args.data, world) or roofline
        if st_serial is not None and st_serial.get("plane_bytes", 0) > 0 and "hbm" in roofline:
            # the same kernel with nothing beside it (the serial pass): what the kernel itself reaches; the figures above are
            # its launches INSIDE the timed region, where the other steps' chains take compute units and bandwidth
            al = sweep_roofline(args, st_serial, nq, args.steps, serial["ms_per_step"], args.data, world)
            if al:
                roofline["alone"] = {k: al[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches")}
                roofline["alone"]["note"] = "the same launches in the `serial` pass (one step at a time, HIP events on the library's stream)"
    elif s16:
        # The bound pass ran on the fp16 matrix cores (k_s16_sweep, csrc/ndbhip_screen16.h).  Algorithmic flops:
        # SURVEY 8d's per-unit figure, 3 x dim per scored (row, query) pair (subtract, multiply, add), x the pairs
        # one launch scores; the kernel EXECUTES 6 x dim per pair (three fp16 products of 2 flops: hi*hi + hi*lo +
        # lo*hi) plus the padding of its 128 x 128 tiles — both against the dense fp16 MFMA peak.
        pairs = bytes_per_launch / (dim * esz)
        alg = 3.0 * dim * pairs
        # pairs whose whole list the triangle inequality excluded before the sweep are algorithmic work the kernel
        # never issued: `executed` counts the rows it did multiply (library_stats.rows_swept, device-counted)
        swept = st.get("rows_swept", 0) / max(1, launches)
        pairs_exe = swept if swept > 0 else pairs
        exe = (6.0 if esz == 4 else 4.0) * dim * pairs_exe               # fp16 rows: two products
        tf = lambda f: f / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
        tr, tr_src = pmc_traffic(args, world, "k_s16_sweep")
        roofline = {"bound": "mfma", "kernel": f"k_s16_sweep<{recipe.replace('R_SCR_', 'R_IVF_')}, {'fp16' if esz == 2 else 'float4'} rows, 4 waves, ring 2>",
                    "achieved": round(tf(alg), 2), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf(alg) / FP16_MFMA_PEAK_TFLOPS, 4),
                    "traffic": tr, "traffic_source": tr_src,
                    "flops_per_launch": int(alg), "avg_launch_ms": round(ms_per_launch, 4), "launches": int(launches),
                    "executed": {"flops_per_launch": int(exe), "achieved": round(tf(exe), 2),
                                 "frac": round(tf(exe) / FP16_MFMA_PEAK_TFLOPS, 4),
                                 "pairs_swept_frac": round(pairs_exe / max(1.0, pairs), 4),
                                 "note": "matrix-core flops actually issued: 6 x dim per (row, query) pair of the lists "
                                         "the sweep multiplied (tile padding not counted; pairs_swept_frac of the "
                                         "algorithmic pairs survive the list- and sublist-level bounds)"},
                    "hbm_traffic": (None if not tr or ms_per_launch <= 0 else
                                    {"achieved": round(tr / (ms_per_launch * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": round(tr / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                     "note": "PMC bytes of one launch (traffic) over this run's launch time: with most "
                                             "(query, sublist) pairs excluded before the sweep, what remains is reading "
                                             "the touched row planes once per batch, so this is the roof that binds"}),
                    "rows_rescored_per_query": round(st.get("rows_rescored", 0) / max(1, nq * args.steps), 1),
                    "rows_emitted_per_query": round(st.get("rows_emitted", 0) / max(1, nq * args.steps), 1),
                    "note": ("bound pass of the screened scan on fp16 matrix cores: rows and queries are split into two fp16 "
                             "planes (x 2^(14-e) = hi + lo), a block multiplies 128 rows x 128 queries from LDS tiles filled "
                             "by LDS DMA, 3 v_mfma_f32_32x32x16_f16 per 16 dimensions; a candidate whose bound cannot "
                             "exclude it is emitted (no distance array), and the reference's sequential arithmetic decides "
                             "among those: ids, ranks and float4 bits are the exact path's.  Before the sweep, the rows of long "
                             "lists are regrouped into sublists (inside the planes only) and a (query, probe) pair expands "
                             "only to the sublists the triangle inequality cannot exclude, so `achieved` (the algorithmic "
                             "flops of ALL scored pairs over the launch time) exceeds `executed` (what was issued).  "
                             "avg_launch_ms = the first (all-queries) sweep; the second sweep for queries that overflowed "
                             "their records is in ms_per_step"),
                    "hbm_algorithmic": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": round(achieved / HBM_PEAK_GBPS, 4),
                                        "note": "rows scored x row bytes per query (SURVEY 8d, no reuse across queries); "
                                                "a staged tile serves 128 queries, the real traffic is `traffic`"}}

    # ---------------- recall@10 vs exact float64 brute force ----------------
    recall = None
    cpu_baseline = None
    if rank == 0 and not sharded and args.rows == "f32" and args.strategy == "l2" and \
            (args.recall_queries > 0 or (args.cpu_seconds > 0 and world == 1)):
        rq = min(args.recall_queries, nq)
        qs = queries[args.warmup * nq: args.warmup * nq + nq]
        step(qs)
        torch.cuda.synchronize()
        got = unpack_tids(out_t[:rq]).cpu().numpy()
        _, _, rows_h, tids_h = ix.export(rows=True)          # index image, list-major (host)
        orig_rows_h = (((tids_h["bi_hi"].astype(np.int64) << 16) | tids_h["bi_lo"]) * 64 + tids_h["posid"] - 1)
        orig_rows = torch.from_numpy(orig_rows_h).to(dev)
        q64 = qs[:rq].double()
        best_d = torch.full((rq, k), float("inf"), dtype=torch.float64, device=dev)
        best_i = torch.zeros((rq, k), dtype=torch.int64, device=dev)
        for s in range(0, n, 65536):
            x = torch.from_numpy(rows_h[s:s + 65536]).to(dev).double()
            d2 = (q64 * q64).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2.0 * (q64 @ x.T)
            dd = torch.cat([best_d, d2], 1)
            ii = torch.cat([best_i, orig_rows[s:s + 65536][None, :].expand(rq, -1)], 1)
            sel = torch.topk(dd, k, dim=1, largest=False)
            best_d, best_i = sel.values, torch.gather(ii, 1, sel.indices)
        gt = best_i.cpu().numpy()
        recall = float(np.mean([len(set(got[i]) & set(gt[i])) / k for i in range(rq)])) if rq > 0 else None

        # ---------------- CPU baseline: the oracle on the host cores (bounded sample) ----------------
        if args.cpu_seconds > 0 and world == 1:
            trace("cpu_baseline")
            cpu_baseline = run_cpu_baseline(args, cent_h, list_len, rows_h, tids_h, qs, out_t, out_d, out_c)

    dist_parity = None
    if sharded and args.dist_parity_queries > 0 and args.rows == "f32" and args.strategy == "l2":
        # every rank runs the step (collectives); rank 0 replays a sample on the CPU oracle
        qs = queries[args.warmup * nq: args.warmup * nq + nq]
        step(qs)
        barrier()
        if rank == 0:
            dist_parity = run_dist_parity(args, full_image, qs, out_t, out_d, out_c)

    # ---------------- replicas: the sharded path all the same, after the timed region ----------------
    # One index sharded over the ranks (heavy lists cut into slices), every batch merged over the library's RCCL
    # communicator: what a table too large for one device needs (BASELINE configs 4 and 5).  Same queries on every
    # rank; rank 0 replays a sample on the CPU oracle.
    sharded_leg = None
    if replicas and args.dist_impl == "c":
        try:
            sh, sh_info = make_shard(ix, "slices")
            qsh = make_data(nq * 6, dim, args.data, args.components, args.sigma, 0x5EED0008, 0x5EEDC0DE, dev)
            sh.search_sharded_device(qsh[:nq], out_t, out_d, out_c, strategy, nprobe, k, 0)        # warm-up
            barrier()
            t0 = time.perf_counter()
            for sidx in range(1, 6):
                sh.search_sharded_device(qsh[sidx * nq:(sidx + 1) * nq], out_t, out_d, out_c, strategy, nprobe, k, 0)
            barrier()
            tsh = time.perf_counter() - t0
            tt = torch.tensor([tsh], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tsh = float(tt.item())
            par = None
            if args.dist_parity_queries > 0 and args.rows == "f32" and args.strategy == "l2":
                sh.search_sharded_device(qsh[:nq], out_t, out_d, out_c, strategy, nprobe, k, 0)
                barrier()
                if rank == 0:
                    par = run_dist_parity(args, full_image, qsh[:nq], out_t, out_d, out_c)
            sh.close()
            sharded_leg = {"what": "the same index cut into list slices over the ranks, every 4096-query batch answered by "
                                   "ndbhip_ivf_search_sharded (query-split selection, two RCCL all-gathers, replay merge): "
                                   "strong scaling of one batch stream",
                           "queries_per_s": round(nq * 5 / tsh, 1), "ms_per_step": round(tsh / 5 * 1e3, 3),
                           "shard": sh_info, "dist_parity_on_sample": par}
        except Exception as e:
            sharded_leg = {"error": f"{type(e).__name__}: {e}"}
    elif replicas and comm_error:
        sharded_leg = {"error": "the library's RCCL communicator could not be opened: " + comm_error}

    gauss = balanced = None
    clustered = None
    if rank == 0 and world == 1 and args.gauss_steps > 0 and args.rows == "f32" and args.strategy == "l2":
        ix.close()
        ix = None
        torch.cuda.empty_cache()
        if args.data == "clustered":
            try:
                trace("iid_gauss leg")
                gauss = gauss_leg(args, dev, steps=args.gauss_steps)
            except Exception as e:
                gauss = {"error": f"{type(e).__name__}: {e}"}
        else:
            # the other table of the line: SURVEY 8d's clustered variant (one component per list, sigma 0.1) through the same
            # step, steps in flight like rounds 5's `value` and one at a time beside it: `value_clustered`
            try:
                trace("clustered leg")
                clustered = l2_table_leg(args, dev, n, dim, nlists, nprobe, k, nq, args.components, args.sigma, steps=max(6, min(args.steps, 20)),
                                         warm=2, nreplay=8, recall_q=64, inflight=lanes_legs, pmc_kind="clustered",
                                         label=f"IVFFlat {n}x{dim} fp32 lists={nlists} probes={nprobe} k={k} L2, {nq} queries/step, "
                                               f"mixture of {args.components} Gaussians, sigma {args.sigma}")
            except Exception as e:
                clustered = {"error": f"{type(e).__name__}: {e}"}
        if args.components == args.lists:
            try:
                torch.cuda.empty_cache()
                trace("balanced_index leg")
                balanced = gauss_leg(args, dev, steps=args.gauss_steps, kind="balanced")
            except Exception as e:
                balanced = {"error": f"{type(e).__name__}: {e}"}

    c5 = None
    c4 = None
    sigma_sweep = None
    if rank == 0 and world == 1 and args.c5_nvec > 0 and args.rows == "f32" and args.strategy == "l2":
        try:
            if ix is not None:
                ix.close()
                ix = None
            torch.cuda.empty_cache()
            trace("c5 leg")
            free_b, _tot = torch.cuda.mem_get_info(dev)
            need_b = 4.5 * args.c5_nvec * 1536 * 4
            if free_b < need_b:
                c5 = {"skipped": f"{free_b / 2**30:.0f} GiB free on the device, the leg needs about {need_b / 2**30:.0f} GiB"}
            else:
                c5 = c5_leg(args, dev, args.c5_nvec)
        except Exception as e:
            c5 = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    default_wl = rank == 0 and world == 1 and args.rows == "f32" and args.strategy == "l2"
    if default_wl and (args.c4_nvec > 0 or args.sigma_sweep):
        if ix is not None:
            ix.close()
            ix = None
        torch.cuda.empty_cache()
    if default_wl and args.c4_nvec > 0:
        try:
            trace("c4 leg")
            free_b, _tot = torch.cuda.mem_get_info(dev)
            need_b = 3.0 * args.c4_nvec * 768 * 4
            if free_b < need_b:
                c4 = {"skipped": f"{free_b / 2**30:.0f} GiB free on the device, the leg needs about {need_b / 2**30:.0f} GiB"}
            else:
                # BASELINE.md C4's table on ONE GPU (it names 8: `--gpus N` shards this same table, and N = 1 of that is this)
                c4 = l2_table_leg(args, dev, args.c4_nvec, 768, 4096, 32, 10, 4096, 4096, 0.1, steps=9, warm=2, nreplay=4, recall_q=32, inflight=args.lanes_legs,
                                  label=f"IVFFlat {args.c4_nvec}x768 fp32 lists=4096 probes=32 k=10 L2, 4096 queries/step, clustered "
                                        f"(4096 components, sigma 0.1), one GPU (BASELINE.md C4 names 8)")
        except Exception as e:
            c4 = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if default_wl and args.sigma_sweep:
        trace("sigma sweep")
        # (sigma 0.1 is the `clustered` leg of the default line: not run twice)
        sigma_sweep = sigma_sweep_leg(args, dev, (0.2, 0.3, 0.5, 1.0) if (clustered and "error" not in clustered and args.sigma == 0.1)
                                      else (0.1, 0.2, 0.3, 0.5, 1.0))
        if clustered and "error" not in clustered and args.sigma == 0.1:
            sigma_sweep = {"sigma_0.1": clustered, **sigma_sweep}

    hnsw = None
    if rank == 0 and world == 1 and args.hnsw_nvec > 0:
        try:
            trace("hnsw leg (ref_compat)")
            hnsw = hnsw_leg(args, dev)
        except Exception as e:                      # the IVF line must not depend on this leg
            hnsw = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        line = {
            "metric": f"kNN queries/sec @ recall@10, {n}x{dim} {'fp32' if esz == 4 else 'fp16'} "
                      f"(IVFFlat lists={nlists} probes={nprobe} k={k} {args.strategy.upper()}; "
                      + ("i.i.d. N(0,1): BASELINE.md section 2's table" if args.data == "gauss" else
                         f"mixture of {args.components} Gaussians sigma {args.sigma}: SURVEY 8d's clustered variant") + ")",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if (replicas or world == 1) else "strong", "vs_baseline": None,
            "dtype": "f32" if esz == 4 else "f32 arithmetic on fp16 rows",
            "data": "synthetic (in-repo counter-based generator csrc/ndbhip_gen.h: ndbhip_gen_rows_device, seeds 0x5EED0001 "
                    "base / 0x5EED0002 queries / 0x5EEDC0DE centers; ndbhip_gen_rows_host regenerates the same bits)",
            "config": {"workload": f"IVFFlat {n}x{dim} {'fp32' if esz == 4 else 'fp16'} lists={nlists} probes={nprobe} "
                                   f"k={k} {args.strategy.upper()}, "
                                   f"{nq} queries/step, exact fp32-sequential arithmetic (bit-identical to the CPU path)",
                       "steps_in_flight_n": inflight,
                       "steps_in_flight": (f"{inflight}: {inflight} handles on ONE mirror of the index (ndbhip_ivf_share), each driven by a host thread on a stream of its "
                                           f"own (ndbhip_set_thread_stream) — a step's per-query chains run under another step's sweep, "
                                           f"the sweeps queue up; `serial` = one step after the other" if inflight > 1 else "1"),
                       "sharding": "none" if not use_dist else
                                   (f"replicas: each of the {world} ranks holds the whole index and answers its own "
                                    f"{nq}-query batches (value = {world} batches per step; no data-path collective); the "
                                    f"sharded path runs after the timed region: sharded_leg") if replicas else
                                   f"{'list slices' if shard_mode == 'slices' else 'whole lists'} over {world} ranks, "
                                   f"balanced by calibration-batch work; RCCL all-gather of probes and records + merge "
                                   f"({'inside the C library: ndbhip_ivf_search_sharded' if args.dist_impl == 'c' else 'torch.distributed calls'})",
                       "shard": shard_info,
                       "collectives": (None if not sharded else
                                       {"rccl_ranks": int(lib().ndbhip_comm_world()) if args.dist_impl == "c" else world,
                                        "per_step": 3,
                                        "what": "all-gather of the probes (the selection is split over the ranks' query ranges), all-reduce "
                                                "(min) of the first thresholds, all-gather of each rank's <= 3k candidates per query",
                                        "bytes_per_step_per_rank": int(nq * nprobe * 4 // world + 2 * nq * 4 + nq * (3 * k * 16 + 12))}),
                       "data": (f"mixture of {args.components} Gaussians, sigma={args.sigma}" if args.data == "clustered"
                                else "i.i.d. N(0,1)"),
                       "index_build": f"ndbhip_ivf_build_device: first-10000-row sample, {kmeans_iters} Lloyd iterations "
                                      f"(reference k-means rule), all rows assigned, {t_build:.3f} s",
                       "list_len_min_mean_max": [int(list_len.min()), float(list_len.mean()), int(list_len.max())],
                       "tables": {"this line (`value`)": (f"mixture of {args.components} Gaussians, sigma={args.sigma} (SURVEY 8d's "
                                                          f"clustered variant): {qps:.0f} queries/s" if args.data == "clustered"
                                                          else f"i.i.d. N(0,1): {qps:.0f} queries/s"),
                                  "i.i.d. N(0,1) (BASELINE.md section 2), same shape, same binary: `iid_gauss`":
                                      (None if not gauss or "queries_per_s" not in gauss else
                                       f"{gauss['queries_per_s']:.0f} queries/s, recall@10 {gauss['recall_at_10']}"),
                                  "clustered variant, same shape, same binary: `clustered`":
                                      (None if not clustered or "queries_per_s" not in clustered else
                                       f"{clustered['queries_per_s']:.0f} queries/s, recall@10 {clustered['recall_at_10']}")}},
            "recall_at_10": None if recall is None else round(recall, 4),
            # the same step on BASELINE.md's own data (i.i.d. N(0,1) rows; the whole leg: `iid_gauss`), at the top level
            # next to `value`: both tables, each with its recall
            # one step from its first launch to its results, whatever else is in flight (with steps in flight
            # ms_per_step is an inverse throughput)
            "step_latency_ms": (serial["ms_per_step"] if serial else round(elapsed / args.steps * 1e3, 3)),
            "value_clustered": None if not clustered or "queries_per_s" not in clustered else clustered["queries_per_s"],
            "ms_per_step_clustered": None if not clustered or "queries_per_s" not in clustered else clustered["ms_per_step"],
            "recall_at_10_clustered": None if not clustered or "queries_per_s" not in clustered else clustered["recall_at_10"],
            "roofline_clustered": None if not clustered or "queries_per_s" not in clustered else clustered.get("roofline"),
            "clustered": clustered,
            "value_iid": None if not gauss or "queries_per_s" not in gauss else gauss["queries_per_s"],
            "recall_at_10_iid": None if not gauss or "recall_at_10" not in gauss else gauss["recall_at_10"],
            "build_vectors_per_s": None if build_vps is None else round(build_vps, 1),
            "build": build,
            "bytes_per_query": int(st["bytes_scored"] / max(1, nq * args.steps)) + nlists * dim * 4,
            "note": None if (args.rows == "f32" and args.strategy == "l2") else
            "recall / CPU legs run for the default fp32 L2 workload only; parity of this variant: tests/test_gpu_ivf.py",
            "roofline": roofline,
            "serial": serial,
            "replicated": replicated,
            "library_stats": {k2: (round(v2, 3) if isinstance(v2, float) else int(v2)) for k2, v2 in st.items()},
            "cpu_baseline": cpu_baseline,
            "dist_parity_on_sample": dist_parity,
            "sharded_leg": sharded_leg,
            "iid_gauss": gauss,
            "balanced_index": balanced,
            "c5": c5,
            "c4": c4,
            "sigma_sweep": sigma_sweep,
            "hnsw": hnsw,
        }
        emit(line, json_fd)
    if use_dist:
        if args.dist_impl == "c":
            check(lib().ndbhip_comm_destroy())
        dist.destroy_process_group()


def run_dist_parity(args, image, qs, out_t, out_d, out_c):
    """N > 1: the merged (all-gather + replay) results of a sample against the oracle's single-process search."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ndbo
    cent_h, list_len, rows_h, tid_h = image
    tid_h = np.ascontiguousarray(tid_h).view(ndbo.TID_DTYPE).reshape(-1)
    off = np.zeros(len(list_len) + 1, dtype=np.int64)
    off[1:] = np.cumsum(list_len)
    img = ndbo.IvfImage(cent_h, off, rows_h, tid_h)
    ns = min(args.dist_parity_queries, len(qs))
    q_h = qs[:ns].cpu().numpy()
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
        res = list(ex.map(lambda i: img.search(q_h[i], 1, args.probes, args.k, 0), range(ns)))
    gt = ndbo.tids_from_device_u64(out_t[:ns].cpu().numpy())
    gd = out_d[:ns].cpu().numpy()
    gc = out_c[:ns].cpu().numpy()
    bad = 0
    for i, (et, ed, _) in enumerate(res):
        bad += not (gc[i] == len(et) and np.array_equal(gt[i, :len(et)], ndbo.tids_to_u64(et)) and
                    np.array_equal(gd[i, :len(et)].view(np.uint32), ed.view(np.uint32)))
    return {"queries": ns, "mismatches": int(bad)}


def gauss_leg(args, dev, steps=3, nrecall=100, nparity=128, kind="gauss"):
    """kind "gauss": the same workload on i.i.d. N(0,1) data (SURVEY 8d's default distribution; reported separately
    because the reference's build rule — k-means on the first 10 000 rows for 1024 centroids — collapses on it: the
    probed lists hold most of the table).  kind "balanced": the headline's clustered table under a BALANCED index —
    the generator's own component centres as centroids (what a k-means that converged would find), every row
    assigned with the insert rule — i.e. the same kernels without the skew the reference's sampling rule puts
    into the lists ("report both").  Build, search, recall@10 vs float64 brute force, oracle parity
    on a sample."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    from oracle import ndbo
    n, dim, nlists, nprobe, k, nq = args.nvec, args.dim, args.lists, args.probes, args.k, args.batch
    ix = IvfIndex(dim, nlists, device=dev.index or 0)
    if kind == "gauss":
        base = make_data(n, dim, "gauss", 1, 0.0, 0x5EED0001, 0, dev)
        q = make_data(nq * (steps + 1), dim, "gauss", 1, 0.0, 0x5EED0002, 0, dev)
        t0 = time.perf_counter()
        iters = ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
        check(lib().ndbhip_synchronize())
        tb = time.perf_counter() - t0
    else:
        base = make_data(n, dim, "clustered", args.components, args.sigma, 0x5EED0001, 0x5EEDC0DE, dev)
        q = make_data(nq * (steps + 1), dim, "clustered", args.components, args.sigma, 0x5EED0002, 0x5EEDC0DE, dev)
        # sigma = 0 rows are their component's centre: one of each
        c0 = make_data(max(200000, 200 * args.components), dim, "clustered", args.components, 0.0, 0x5EED0009, 0x5EEDC0DE, dev)
        key = c0[:, 0].contiguous()
        uniq, inv = torch.unique(key, return_inverse=True)
        first = torch.full((len(uniq),), len(key), dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, inv, torch.arange(len(key), device=dev), reduce="amin")
        cents = c0[first].contiguous()
        del c0
        if len(cents) != nlists:
            raise RuntimeError(f"{len(cents)} distinct centres for {nlists} lists")
        t0 = time.perf_counter()
        ix.set_centroids(cents.cpu().numpy())
        asg = torch.empty(n, dtype=torch.int32, device=dev)
        check(lib().ndbhip_ivf_assign_device(C.c_void_p(cents.data_ptr()), nlists, dim, C.c_void_p(base.data_ptr()), n,
                                            C.c_void_p(asg.data_ptr())))
        check(lib().ndbhip_synchronize())
        order = torch.argsort(asg.long(), stable=True)
        ll = torch.bincount(asg.long(), minlength=nlists).cpu().numpy()
        rows_sorted = base[order].contiguous()
        ix.load_device(ll, rows_sorted, pack_tids(order))
        check(lib().ndbhip_synchronize())
        tb = time.perf_counter() - t0
        iters = 0
        base = rows_sorted                      # (recall below needs row ids: recover them from the TIDs' order)
        row_id = order
    ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    ix.search_device(q[:nq], ot, od, oc, 1, nprobe, k, 0)          # warm-up (planes, workspaces)
    torch.cuda.synchronize()
    check(lib().ndbhip_stats_reset())
    check(lib().ndbhip_profile(1))
    t0 = time.perf_counter()
    for sidx in range(steps):
        ix.search_device(q[(sidx + 1) * nq:(sidx + 2) * nq], ot, od, oc, 1, nprobe, k, 0)
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / steps
    check(lib().ndbhip_profile(0))
    st = _lib.stats()
    qs = q[steps * nq:(steps + 1) * nq]
    got = unpack_tids(ot[:nrecall]).cpu().numpy()
    q64 = qs[:nrecall].double()
    best_d = torch.full((nrecall, k), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.zeros((nrecall, k), dtype=torch.int64, device=dev)
    for s0 in range(0, n, 65536):
        x = base[s0:s0 + 65536].double()
        d2 = (q64 * q64).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2.0 * (q64 @ x.T)
        dd = torch.cat([best_d, d2], 1)
        ii = torch.cat([best_i, torch.arange(s0, s0 + len(x), device=dev)[None, :].expand(nrecall, -1)], 1)
        sel = torch.topk(dd, k, dim=1, largest=False)
        best_d, best_i = sel.values, torch.gather(ii, 1, sel.indices)
    if kind != "gauss":
        best_i = row_id[best_i]                 # brute force ran over the list-major copy
    gt = best_i.cpu().numpy()
    recall = float(np.mean([len(set(got[i]) & set(gt[i])) / k for i in range(nrecall)]))
    del base
    cent_h, list_len, rows_h, tid_h = ix.export(rows=True)
    tid_h = np.ascontiguousarray(tid_h).view(ndbo.TID_DTYPE).reshape(-1)
    off = np.zeros(len(list_len) + 1, dtype=np.int64)
    off[1:] = np.cumsum(list_len)
    img = ndbo.IvfImage(cent_h, off, rows_h, tid_h)
    q_h = qs[:nparity].cpu().numpy()
    cores = host_cores()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(lambda i: img.search(q_h[i], 1, nprobe, k, 0), range(nparity)))
    t_cpu = time.perf_counter() - t0
    gtid = ndbo.tids_from_device_u64(ot[:nparity].cpu().numpy())
    gd, gc = od[:nparity].cpu().numpy(), oc[:nparity].cpu().numpy()
    bad = 0
    for i, (et, ed, _) in enumerate(res):
        bad += not (gc[i] == len(et) and np.array_equal(gtid[i, :len(et)], ndbo.tids_to_u64(et)) and
                    np.array_equal(gd[i, :len(et)].view(np.uint32), ed.view(np.uint32)))
    ix.close()
    what = "i.i.d. N(0,1)" if kind == "gauss" else \
        "the headline's clustered table, centroids = the generator's component centres (balanced lists)"
    return {"workload": f"IVFFlat {n}x{dim} fp32 lists={nlists} probes={nprobe} k={k} L2, {nq} queries/step, {what}",
            "queries_per_s": round(nq / ts, 1), "ms_per_step": round(ts * 1e3, 3), "recall_at_10": round(recall, 4),
            "build_vectors_per_s": round(n / tb, 1), "kmeans_iterations": int(iters),
            "bytes_per_query": int(st["bytes_scored"] / max(1, nq * steps)) + nlists * dim * 4,
            "list_len_min_mean_max": [int(list_len.min()), float(list_len.mean()), int(list_len.max())],
            "rows_rescored_per_query": round(st.get("rows_rescored", 0) / max(1, nq * steps), 1),
            "screen16": {"batches": int(st.get("screen16_batches", 0)), "fallbacks": int(st.get("screen16_fallbacks", 0)),
                         "pairs_pruned_frac": round(st.get("pairs_pruned", 0) / max(1, nq * steps * nprobe), 4),
                         "rows_swept_frac": round(st.get("rows_swept", 0) / max(1, st.get("rows_scored", 1)), 4)},
            "roofline": sweep_roofline(args, st, nq, steps, ts * 1e3, "gauss" if kind == "gauss" else "balanced", 1)
            if st.get("plane_bytes", 0) > 0 else None,
            "cpu_baseline": {"value": round(nparity / t_cpu, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                             "sample": f"{nparity} of the step's queries through the C oracle (gcc -O2, the reference's "
                                       f"default flags), one thread per host core; the same sample checks the GPU results"},
            "oracle_parity": {"queries": nparity, "mismatches": int(bad)}}


def steps_in_flight(ix, nlanes, q, nq, first, count, strategy, P, K, dev, warm=2):
    """`count` batches (q[(first + s) * nq : ...], s = 0 .. count - 1) with `nlanes` of them in flight: the source handle and
    nlanes - 1 shares of it (ndbhip_ivf_share: the same rows, planes and tables, scratch of their own), a host thread and a
    stream each (ndbhip_set_thread_stream).  Returns (seconds per step, the last batch's results per lane as
    [(step, tids, dist bits, counts)]).  The source must have run a batch of this kind already (a share builds nothing)."""
    import threading
    from neurondb_amd._lib import check, lib
    handles = [ix] + [ix.share() for _ in range(nlanes - 1)]
    streams = [torch.cuda.Stream() for _ in range(nlanes)]
    bufs = [(torch.zeros((nq, K), dtype=torch.int64, device=dev), torch.zeros((nq, K), dtype=torch.float32, device=dev),
             torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in range(nlanes)]
    err, last = [], [None] * nlanes

    def run(lo, n_):
        def lane(w):
            try:
                check(lib().ndbhip_set_thread_stream(streams[w].cuda_stream))
                for s_ in range(lo + w, lo + n_, nlanes):
                    handles[w].search_device(q[s_ * nq:(s_ + 1) * nq], *bufs[w], strategy, P, K, 0)
                    last[w] = s_
                check(lib().ndbhip_synchronize())
            except Exception as e:          # noqa: BLE001
                err.append(e)
            finally:
                lib().ndbhip_set_thread_stream(None)
        th = [threading.Thread(target=lane, args=(w,)) for w in range(nlanes)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        if err:
            raise err[0]

    try:
        run(first - warm * nlanes if first >= warm * nlanes else first, min(warm * nlanes, count))      # (each lane's scratch grows here)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(first, count)
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / count
        out = [(last[w], bufs[w][0].cpu().numpy().copy(), bufs[w][1].cpu().numpy().view(np.uint32).copy(), bufs[w][2].cpu().numpy().copy())
               for w in range(nlanes) if last[w] is not None]
    finally:
        for h in handles[1:]:
            h.close()
    return ts, out


def l2_table_leg(args, dev, n, dim, lists, P, K, nq, components, sigma, steps=5, warm=2, nreplay=8, recall_q=64, aniso=False,
                 kind="clustered", label="", inflight=1, pmc_kind="leg"):
    """One float4 L2 table of another shape or another spread through the same timed step: build on the device, prepare,
    `steps` batches of `nq` queries, then recall@K of `recall_q` queries against a float64 brute force over the rows, which
    sweep ran, how many (row, pair) elements the bounds excluded, and the CPU oracle's answers for `nreplay` queries over
    their probed lists (ids + float4 bits).  aniso: every dimension d of rows and queries scaled by (d + 1)^-0.5
    (a power-law spectrum: what learned embeddings look like next to an isotropic Gaussian)."""
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    from oracle import ndbo
    base = make_data(n, dim, kind, components, sigma, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq * (steps + warm + 1), dim, kind, components, sigma, 0x5EED0002, 0x5EEDC0DE, dev)
    if aniso:
        w = (torch.arange(1, dim + 1, device=dev, dtype=torch.float32) ** -0.5)[None, :]
        base.mul_(w)
        q.mul_(w)
    ix = IvfIndex(dim, lists, device=dev.index or 0)
    t0 = time.perf_counter()
    iters = ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    ix.prepare(1)
    check(lib().ndbhip_synchronize())
    tb = time.perf_counter() - t0
    ot = torch.zeros((nq, K), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, K), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    for w_ in range(warm):
        ix.search_device(q[w_ * nq:(w_ + 1) * nq], ot, od, oc, 1, P, K, 0)
    torch.cuda.synchronize()
    check(lib().ndbhip_stats_reset())
    check(lib().ndbhip_profile(1))
    t0 = time.perf_counter()
    for sidx in range(steps):
        ix.search_device(q[(warm + sidx) * nq:(warm + sidx + 1) * nq], ot, od, oc, 1, P, K, 0)
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / steps
    check(lib().ndbhip_profile(0))
    st = _lib.stats()
    flight = None
    if inflight > 1:
        # the same steps with `inflight` of them in flight on shares of this mirror; every lane's last batch must be what
        # the serial loop gives for that batch
        tf, lasts = steps_in_flight(ix, inflight, q, nq, warm, steps, 1, P, K, dev)
        same = True
        for s_, lt, ld, lc in lasts:
            ix.search_device(q[s_ * nq:(s_ + 1) * nq], ot, od, oc, 1, P, K, 0)
            torch.cuda.synchronize()
            same = same and np.array_equal(ot.cpu().numpy(), lt) and np.array_equal(od.cpu().numpy().view(np.uint32), ld) and \
                np.array_equal(oc.cpu().numpy(), lc)
        flight = {"steps_in_flight": inflight, "queries_per_s": round(nq / tf, 1), "ms_per_step": round(tf * 1e3, 3),
                  "identical_to_serial": bool(same),
                  "how": "ndbhip_ivf_share: handles on ONE mirror, a host thread and a stream each"}
    # (what the leg reports is the faster of the two ways to run it: where the sweep is bound by the matrix cores and fills
    # the device — the tables whose clusters have blurred — steps in flight only get in each other's way)
    use_flight = bool(flight) and flight["ms_per_step"] < ts * 1e3
    qs = q[(warm + steps) * nq:(warm + steps + 1) * nq]
    ix.search_device(qs, ot, od, oc, 1, P, K, 0)
    torch.cuda.synchronize()
    got_t, got_d, got_c = ot.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy(), oc.cpu().numpy().copy()
    # recall@K: float64 brute force over the table as it lies in `base` (row r has TID r)
    rq = min(recall_q, nq)
    q64 = qs[:rq].double()
    best_d = torch.full((rq, K), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.zeros((rq, K), dtype=torch.int64, device=dev)
    for s0 in range(0, n, 131072):
        x = base[s0:s0 + 131072].double()
        d2 = (q64 * q64).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2.0 * (q64 @ x.T)
        dd = torch.cat([best_d, d2], 1)
        ii = torch.cat([best_i, torch.arange(s0, s0 + x.shape[0], device=dev)[None, :].expand(rq, -1)], 1)
        sel = torch.topk(dd, K, dim=1, largest=False)
        best_d, best_i = sel.values, torch.gather(ii, 1, sel.indices)
    gt = best_i.cpu().numpy()
    got_rows = unpack_tids(ot[:rq]).cpu().numpy()
    recall = float(np.mean([len(set(got_rows[i][:got_c[i]]) & set(gt[i])) / K for i in range(rq)]))
    # the CPU oracle over the probed lists of a few queries
    cent_h, ll, _, _ = ix.export(rows=False)
    t6 = np.zeros((n, 6), np.uint8)
    check(lib().ndbhip_ivf_export(ix._h, None, None, None, t6.ctypes.data_as(C.c_void_p)))
    tid_all = t6.view(ndbo.TID_DTYPE).reshape(n)
    order = (((tid_all["bi_hi"].astype(np.int64) << 16) | tid_all["bi_lo"]) * 64 + tid_all["posid"] - 1)
    off = np.zeros(len(ll) + 1, np.int64)
    off[1:] = np.cumsum(ll)
    qh = qs.cpu().numpy()
    gtid = ndbo.tids_from_device_u64(got_t)
    bad_oracle = 0
    for i in range(nreplay):
        pr = sorted(int(x) for x in ix.select_clusters(qh[i:i + 1], P)[0] if x >= 0)
        keep = np.zeros(len(ll), bool)
        keep[pr] = True
        off2 = np.zeros(len(ll) + 1, np.int64)
        off2[1:] = np.cumsum(np.where(keep, ll, 0))
        sel = np.concatenate([np.arange(off[L], off[L + 1]) for L in pr]) if pr else np.zeros(0, np.int64)
        rows_img = base[torch.from_numpy(order[sel]).to(dev)].cpu().numpy()
        img = ndbo.IvfImage(cent_h, off2, rows_img, np.ascontiguousarray(tid_all[sel]))
        et, ed, _ = img.search(qh[i], 1, P, K, 0)
        bad_oracle += not (got_c[i] == len(et) and np.array_equal(gtid[i, :len(et)], ndbo.tids_to_u64(et)) and
                           np.array_equal(got_d[i, :len(et)], ed.view(np.uint32)))
    ix.close()
    del base, q
    torch.cuda.empty_cache()
    launches = max(1, st["scan_launches"])
    which = ("k_s16c_dense" if st.get("dense_sweeps", 0) > 0 else
             ("centred sweep (k_s16c_wsweep / k_s16c_sweep)" if st.get("plane_bytes", 0) > 0 else
              ("k_s16_sweep" if st.get("screen16_batches", 0) > 0 else "exact / fp32-screened scans")))
    import copy
    a2 = copy.copy(args)
    a2.dim, a2.nvec, a2.lists, a2.batch, a2.probes, a2.k = dim, n, lists, nq, P, K
    return {"workload": label or f"IVFFlat {n}x{dim} fp32 lists={lists} probes={P} k={K} L2, {nq} queries/step",
            "queries_per_s": flight["queries_per_s"] if use_flight else round(nq / ts, 1),
            "ms_per_step": flight["ms_per_step"] if use_flight else round(ts * 1e3, 3), "steps": steps,
            "mode": "steps in flight" if use_flight else "one step at a time",
            "in_flight": flight,
            "serial": {"queries_per_s": round(nq / ts, 1), "ms_per_step": round(ts * 1e3, 3)} if flight else None,
            "recall_at_10": round(recall, 4), "kmeans_iterations": int(iters), "build_and_prepare_s": round(tb, 3),
            "lists_nonempty": int((ll > 0).sum()), "list_len_max": int(ll.max()),
            "sweep": which, "sweep_ms": round(st["scan_kernel_ms"] / launches, 4),
            "pairs_pruned_frac": round(1.0 - st.get("rows_swept", 0) / max(1, st.get("rows_scored", 1)), 4),
            "screen16": {"batches": int(st.get("screen16_batches", 0)), "fallbacks": int(st.get("screen16_fallbacks", 0))},
            "rows_emitted_per_query": round(st.get("rows_emitted", 0) / max(1, nq * steps), 1),
            "rows_rescored_per_query": round(st.get("rows_rescored", 0) / max(1, nq * steps), 1),
            "roofline": sweep_roofline(a2, st, nq, steps, ts * 1e3, pmc_kind, 1) if st.get("plane_bytes", 0) > 0 else None,
            "oracle_parity": {"queries": nreplay, "mismatches": int(bad_oracle),
                              "note": "oracle/ndb_oracle.c over each query's probed lists: TIDs, float4 bits, counts"}}


def sigma_sweep_leg(args, dev, sigmas=(0.1, 0.2, 0.3, 0.5, 1.0)):
    """the regime between the two tables of the line.  The headline shape (1M x 768, lists 1024, probes 32,
    k 10, 4096 queries a step) with the mixture's spread sigma from 0.1 (components apart: the headline) to 1.0 (components
    as wide as their centres are apart: nothing left of them in 768 dimensions), plus one anisotropic table (sigma 0.3, every
    dimension d scaled by (d + 1)^-0.5).  Per table: queries/s (steps in flight like the line's `value`; `serial` beside it),
    recall@10, the share of (row, pair) elements the bounds excluded, which sweep ran, oracle parity."""
    out = {}
    for sg in sigmas:
        try:
            out[f"sigma_{sg}"] = l2_table_leg(args, dev, args.nvec, args.dim, args.lists, args.probes, args.k, args.batch,
                                              args.components, sg, steps=6, warm=2, nreplay=4, recall_q=32, inflight=args.lanes_legs,
                                              label=f"mixture of {args.components} Gaussians, sigma {sg}")
        except Exception as e:
            out[f"sigma_{sg}"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["anisotropic_sigma_0.3"] = l2_table_leg(args, dev, args.nvec, args.dim, args.lists, args.probes, args.k, args.batch,
                                                    args.components, 0.3, steps=6, warm=2, nreplay=4, recall_q=32, aniso=True, inflight=args.lanes_legs,
                                                    label=f"mixture of {args.components} Gaussians, sigma 0.3, dimension d scaled by (d+1)^-0.5")
    except Exception as e:
        out["anisotropic_sigma_0.3"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def c5_leg(args, dev, n, steps=24, warm=3, nreplay=8):
    """BASELINE.md's C5 on one GPU: 10M x 1536 halfvec rows, inner product, lists 4096, probes 32, k 10, batches of 256
    queries (the clustered generator with one component per list).  Build on the device, halfvec twin (round to
    nearest even), timed batches on the screened path, then parity two ways: the whole last batch against the library's
    exact scan (scan mode 3: the kernels the small-size oracle tests pin), and `nreplay` of its queries against the C
    oracle over their probed lists (the other lists emptied: the oracle then collects exactly the candidates the full
    image would give it, without 61 GB of rows crossing to the host)."""
    import copy
    import ctypes as C
    from neurondb_amd import IvfIndex, _lib
    from neurondb_amd._lib import check, lib
    from oracle import ndbo
    dim, lists, nq, K, P, strategy = 1536, 4096, 256, 10, 32, 3
    base = make_data(n, dim, "clustered", lists, 0.1, 0x5EED0001, 0x5EEDC0DE, dev)
    q = make_data(nq * (steps + warm + 1), dim, "clustered", lists, 0.1, 0x5EED0002, 0x5EEDC0DE, dev)
    ix = IvfIndex(dim, lists, device=dev.index or 0)
    t0 = time.perf_counter()
    iters = ix.build_device(base, pack_tids(torch.arange(n, device=dev)), 50)
    check(lib().ndbhip_synchronize())
    tb = time.perf_counter() - t0
    twin = ix.to_f16(False)
    ix.close()
    ix = twin
    ot = torch.zeros((nq, K), dtype=torch.int64, device=dev)
    od = torch.zeros((nq, K), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    for w in range(warm):
        ix.search_device(q[w * nq:(w + 1) * nq], ot, od, oc, strategy, P, K, 0)
    torch.cuda.synchronize()
    check(lib().ndbhip_stats_reset())
    check(lib().ndbhip_profile(1))
    t0 = time.perf_counter()
    for sidx in range(steps):
        ix.search_device(q[(warm + sidx) * nq:(warm + sidx + 1) * nq], ot, od, oc, strategy, P, K, 0)
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / steps
    check(lib().ndbhip_profile(0))
    st = _lib.stats()
    flight = None
    # (batches of 256: a step is a chain of small kernels at their latency floor — twice the lanes of the 4096-query steps)
    nfl = 2 * args.lanes_legs if getattr(args, "lanes_legs", 1) > 1 else 1
    if nfl > 1:
        tf, lasts = steps_in_flight(ix, nfl, q, nq, warm, steps, strategy, P, K, dev)
        same = True
        for s_, lt, ld, lc in lasts:
            ix.search_device(q[s_ * nq:(s_ + 1) * nq], ot, od, oc, strategy, P, K, 0)
            torch.cuda.synchronize()
            same = same and np.array_equal(ot.cpu().numpy(), lt) and np.array_equal(od.cpu().numpy().view(np.uint32), ld) and \
                np.array_equal(oc.cpu().numpy(), lc)
        flight = {"steps_in_flight": nfl, "queries_per_s": round(nq / tf, 1), "ms_per_step": round(tf * 1e3, 3),
                  "identical_to_serial": bool(same),
                  "how": "ndbhip_ivf_share: handles on ONE mirror, a host thread and a stream each"}
    # parity 1: the last batch again, screened vs the exact scan
    qs = q[(warm + steps) * nq:(warm + steps + 1) * nq]
    ix.search_device(qs, ot, od, oc, strategy, P, K, 0)
    torch.cuda.synchronize()
    got = (ot.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy(), oc.cpu().numpy().copy())
    check(lib().ndbhip_set_scan_mode(3))
    try:
        ix.search_device(qs, ot, od, oc, strategy, P, K, 0)
        torch.cuda.synchronize()
    finally:
        check(lib().ndbhip_set_scan_mode(0))
    exact = (ot.cpu().numpy(), od.cpu().numpy().view(np.uint32), oc.cpu().numpy())
    bad_exact = int(sum(not (got[2][i] == exact[2][i] and np.array_equal(got[0][i], exact[0][i]) and
                             np.array_equal(got[1][i], exact[1][i])) for i in range(nq)))
    # parity 2: the oracle over the probed lists of a few queries (rows as the reference decodes the halfvec values)
    cent_h, ll, _, _ = ix.export(rows=False)
    t6 = np.zeros((n, 6), np.uint8)
    check(lib().ndbhip_ivf_export(ix._h, None, None, None, t6.ctypes.data_as(C.c_void_p)))
    tid_all = t6.view(ndbo.TID_DTYPE).reshape(n)
    order = (((tid_all["bi_hi"].astype(np.int64) << 16) | tid_all["bi_lo"]) * 64 + tid_all["posid"] - 1)
    off = np.zeros(len(ll) + 1, np.int64)
    off[1:] = np.cumsum(ll)
    qh = qs.cpu().numpy()
    gtid = ndbo.tids_from_device_u64(got[0])
    bad_oracle = 0
    cores = host_cores()
    t_cpu = 0.0
    for i in range(nreplay):
        pr = sorted(int(x) for x in ix.select_clusters(qh[i:i + 1], P)[0] if x >= 0)
        keep = np.zeros(len(ll), bool)
        keep[pr] = True
        off2 = np.zeros(len(ll) + 1, np.int64)
        off2[1:] = np.cumsum(np.where(keep, ll, 0))
        sel = np.concatenate([np.arange(off[L], off[L + 1]) for L in pr])
        h = base[torch.from_numpy(order[sel]).to(dev)].to(torch.float16)
        f = h.to(torch.float32)
        rows_img = torch.where((h.abs() < 2.0 ** -14) & (h != 0), f * 2.0 ** -10, f).cpu().numpy()     # quirk Q20
        img = ndbo.IvfImage(cent_h, off2, rows_img, np.ascontiguousarray(tid_all[sel]))
        t0 = time.perf_counter()
        et, ed, _ = img.search(qh[i], strategy, P, K, 0)
        t_cpu += time.perf_counter() - t0
        bad_oracle += not (got[2][i] == len(et) and np.array_equal(gtid[i, :len(et)], ndbo.tids_to_u64(et)) and
                           np.array_equal(got[1][i, :len(et)], ed.view(np.uint32)))
    ix.close()
    a5 = copy.copy(args)
    a5.dim, a5.strategy, a5.rows, a5.nvec, a5.lists, a5.batch, a5.probes, a5.k = dim, "ip", "f16", n, lists, nq, P, K
    return {"workload": f"IVFFlat {n}x{dim} halfvec lists={lists} probes={P} k={K} inner product, {nq} queries/step, "
                        f"clustered ({lists} components, sigma 0.1), one GPU (BASELINE.md C5 names 8)",
            "queries_per_s": flight["queries_per_s"] if flight else round(nq / ts, 1),
            "ms_per_step": flight["ms_per_step"] if flight else round(ts * 1e3, 3), "steps": steps,
            "in_flight": flight,
            "serial": {"queries_per_s": round(nq / ts, 1), "ms_per_step": round(ts * 1e3, 3)} if flight else None,
            "build_vectors_per_s": round(n / tb, 1), "kmeans_iterations": int(iters),
            "list_len_min_mean_max": [int(ll.min()), float(ll.mean()), int(ll.max())],
            "screen16": {"batches": int(st.get("screen16_batches", 0)), "fallbacks": int(st.get("screen16_fallbacks", 0)),
                         "rows_swept_frac": round(st.get("rows_swept", 0) / max(1, st.get("rows_scored", 1)), 4)},
            "roofline": sweep_roofline(a5, st, nq, steps, ts * 1e3, "c5", 1) if st.get("plane_bytes", 0) > 0 else None,
            "exact_scan_parity": {"queries": nq, "mismatches": bad_exact,
                                  "note": "the screened batch against scan mode 3 (exact kernels) on the same mirror: TIDs, "
                                          "float4 bits, counts"},
            "oracle_parity": {"queries": nreplay, "mismatches": int(bad_oracle),
                              "note": "the C oracle over each query's probed lists (rows decoded like fp16_to_float)"},
            "cpu_baseline": {"value": round(nreplay / t_cpu, 2) if t_cpu > 0 else None, "unit": "queries/s", "cores": 1,
                             "kind": "port", "sample": f"the {nreplay} replayed queries, one at a time on one core"}}


def hnsw_leg(args, dev, m=16, efc=200, ef=64, nq=8192):
    """BASELINE config C3: HNSW m=16 ef_search=64 k=10 cosine on unit-norm rows — device build
    (ndbhip_hnsw_build_device) and batch search, with a sample of the queries replayed by the CPU oracle
    on the exported graph (blocks, ranks, float4 bits and evaluation counts must all agree)."""
    import ctypes as C
    from neurondb_amd import HnswIndex
    from neurondb_amd._lib import check, lib
    from oracle import ndbo
    n, dim, k = args.hnsw_nvec, args.dim, args.k
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0003)
    base = torch.randn((n, dim), generator=g, device=dev)
    base = base / base.norm(dim=1, keepdim=True)
    q = torch.randn((nq, dim), generator=g, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
    levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)      # hnsw_am.c:1143-1161
    tids = pack_tids(torch.arange(n, device=dev))
    ix = HnswIndex(dim, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib().ndbhip_hnsw_build_device(ix._h, C.c_void_p(base.data_ptr()), C.c_void_p(tids.data_ptr()), n,
                                         levels.ctypes.data, efc))
    check(lib().ndbhip_synchronize())
    tb = time.perf_counter() - t0
    ix.nblocks = n + 1
    ob = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    od = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    oc = torch.zeros(nq, dtype=torch.int32, device=dev)
    ot = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    osc = torch.zeros(nq, dtype=torch.int64, device=dev)

    def run():
        check(lib().ndbhip_hnsw_search_device(ix._h, C.c_void_p(q.data_ptr()), nq, 2, ef, k,
                                              C.c_void_p(ob.data_ptr()), C.c_void_p(od.data_ptr()),
                                              C.c_void_p(oc.data_ptr()), C.c_void_p(ot.data_ptr()),
                                              C.c_void_p(osc.data_ptr())))
        check(lib().ndbhip_synchronize())
    run()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ts = (time.perf_counter() - t0) / reps
    evals = float(osc.double().mean())
    bytes_q = evals * (48 + 4 * dim + 4 * 2 * m)            # E x (node header + vector + level-0 slots)
    sims = q[:200].double() @ base.double().T
    gt = torch.topk(sims, k, dim=1).indices.cpu().numpy() + 1
    got = ob.cpu().numpy()
    cnt = oc.cpu().numpy()
    recall = float(np.mean([len(set(got[i][:cnt[i]]) & set(gt[i])) / k for i in range(200)]))
    # oracle replay on the device-built graph
    e = ix.export()
    vecs = np.zeros((n + 1, dim), np.float32)
    vecs[1:] = base.cpu().numpy()
    og = ndbo.HnswGraph.from_arrays(vecs, e["levels"], e["ncount"], e["nbrs"], None, e["entry_point"],
                                    e["entry_level"], m, efc)
    sample, bad = 32, 0
    qh, gd, gs = q[:sample].cpu().numpy(), od.cpu().numpy(), osc.cpu().numpy()
    t0 = time.perf_counter()
    for i in range(sample):
        eb, ed, ns = og.search(qh[i], 2, ef, k)
        bad += not (cnt[i] == len(eb) and np.array_equal(got[i, :len(eb)], eb) and
                    np.array_equal(gd[i, :len(eb)].view(np.uint32), ed.view(np.uint32)) and gs[i] == ns)
    tc = (time.perf_counter() - t0) / sample
    # all host cores (one thread per core; the ctypes call releases the GIL): the CPU baseline of C3 in queries/s
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()
    ncpu = int(min(nq, max(cores * 8, min(args.cpu_seconds, 10.0) * cores / max(tc, 1e-6))))
    qall = q[:ncpu].cpu().numpy()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda lo: [og.search(qall[i], 2, ef, k) for i in range(lo, min(ncpu, lo + 16))], range(0, ncpu, 16)))
    cpu_qps = ncpu / (time.perf_counter() - t0)

    # the reference's unused best-first search (src/scan/hnsw_scan.c) on the same graph and queries
    # (always compute_l2_distance; on unit-norm rows L2 and cosine rank alike, so the same ground truth serves)
    def run_layer():
        check(lib().ndbhip_hnsw_search_layer_device(ix._h, C.c_void_p(q.data_ptr()), nq, 1, ef, k,
                                                    C.c_void_p(ob.data_ptr()), C.c_void_p(od.data_ptr()),
                                                    C.c_void_p(oc.data_ptr()), C.c_void_p(ot.data_ptr()),
                                                    C.c_void_p(osc.data_ptr())))
        check(lib().ndbhip_synchronize())
    run_layer()
    t0 = time.perf_counter()
    for _ in range(3):
        run_layer()
    tl = (time.perf_counter() - t0) / 3
    got_l, cnt_l, gd_l, gs_l = ob.cpu().numpy(), oc.cpu().numpy(), od.cpu().numpy(), osc.cpu().numpy()
    recall_l = float(np.mean([len(set(got_l[i][:cnt_l[i]]) & set(gt[i])) / k for i in range(200)]))
    bad_l = 0
    for i in range(sample):
        eb, ed, ns = og.search_layer(qh[i], ef, k)
        bad_l += not (cnt_l[i] == len(eb) and np.array_equal(got_l[i, :len(eb)], eb) and
                      np.array_equal(gd_l[i, :len(eb)].view(np.uint32), ed.view(np.uint32)) and gs_l[i] == ns)
    layer = {"what": "hnsw_search_layer (src/scan/hnsw_scan.c, the reference's unused best-first search), "
                     f"ef={ef} k={k}, same graph and queries",
             "queries_per_s": round(nq / tl, 1), "ms_per_batch": round(tl * 1e3, 3),
             "evaluations_per_query": round(float(gs_l.mean()), 1), "recall_at_10": round(recall_l, 4),
             "oracle_parity": {"queries": sample, "mismatches": int(bad_l),
                               "checked": "blocks in slot order, float4 bits, evaluation counts"}}
    trace("hnsw leg (intended)")
    intended = hnsw_intended_leg(args, dev)
    ref_compat = {"what": "hnswSearch as the reference runs it (hnsw_am.c:1545-2080: greedy descent, BFS-until-ef at level 0, "
                          "quirks Q10 / Q12): PARITY evidence — blocks, ranks, float4 bits and evaluation counts equal the "
                          "oracle's — not a search result anybody wants (recall ~ 0 by the reference's own algorithm)",
                  "build_vectors_per_s": round(n / tb, 1), "build_s": round(tb, 3), "build_schedule": ix.build_stats(),
                  "queries_per_s": round(nq / ts, 1), "ms_per_batch": round(ts * 1e3, 3),
            "evaluations_per_query": round(evals, 1), "bytes_per_query": int(bytes_q),
            "roofline": hnsw_roofline(n, dim, m, ef, nq, ts, bytes_q),
            "recall_at_10": round(recall, 4),
            "recall_note": "the reference's level-0 walk is BFS-until-ef (quirk Q10); the oracle returns the same ids",
            "oracle_parity": {"queries": sample, "mismatches": int(bad),
                              "checked": "blocks, ranks, float4 bits, evaluation counts"},
            "cpu_oracle_ms_per_query_single_thread": round(tc * 1e3, 3),
            "cpu_baseline": {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                             "sample": f"{ncpu} of the same queries through oracle/ndb_oracle.c ndbo_hnsw_search on the "
                                       "exported graph, one thread per core"},
            "search_layer": layer}
    # C3's figure is the `intended` index (the layer search src/scan/hnsw_scan.c specifies, on a graph
    # built with the links hnswInsertNode drops): queries_per_s / recall_at_10 at the top are ITS
    top = {"workload": f"HNSW {n}x{dim} fp32 m={m} ef_construction={efc} ef_search={ef} k={k} cosine, "
                       f"{nq}-query batches (BASELINE config C3)"}
    if intended and "queries_per_s" in intended:
        for key in ("queries_per_s", "recall_at_10", "build_vectors_per_s", "ms_per_batch", "evaluations_per_query", "roofline",
                    "in_flight", "one_batch_at_a_time", "strategy"):
            if key in intended:
                top[key] = intended[key]
        top["mode"] = "intended (details: `intended`); the reference-compatible walk: `ref_compat`"
    top["intended"] = intended
    top["ref_compat"] = ref_compat
    return top


def hnsw_roofline(n, dim, m, ef, nq, ts, bytes_q):
    """The reference-compatible walk is a chain of dependent fetches around the entry point: what reaches HBM is the
    PMC traffic of a committed pass (frac = that over this run's batch time over 8 TB/s, about 0.01), the rest is
    served from L2; the algorithmic bytes SURVEY 8d defines (evaluations x node bytes) are kept beside it."""
    t = hnsw_pmc_traffic(n, dim, m, ef, nq, ts)
    tr = t.get("traffic")
    return {"bound": "latency", "achieved": None if not tr else round(tr / ts / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": t.get("traffic_frac"), "traffic": tr, "traffic_source": t.get("traffic_source"),
            "algorithmic": {"bytes_per_query": int(bytes_q), "gb_per_s": round(nq / ts * bytes_q / 1e9, 1),
                            "note": "evaluations x (node header + vector + level-0 slots), SURVEY 8d; every query walks the "
                                    "same ~100 nodes around the entry point, so these bytes come from L2, not HBM"},
            "note": "dependent graph walk, latency-bound; frac = HBM-side PMC bytes per launch / batch time / 8 TB/s"}


def hnsw_intended_leg(args, dev, m=16, efc=200, ef=64, nq=8192, strategy=2):
    """BASELINE config C3 in the `intended` mode (include/ndbhip.h ndbhip_hnsw_build_intended_device,
    oracle/ndb_oracle_hnsw2.c): build and search in HBM, recall@10 against a float64 brute force, a sample replayed by
    the oracle on the exported graph (blocks, float4 distance bits, evaluation counts).  Table: the headline's clustered
    generator, rows normalised (cosine order = L2 order); the i.i.d. N(0,1) table is measured beside it at 100 000
    rows — in 768 dimensions its nearest neighbours are barely nearer than the rest and no graph walk of ef = 64
    finds them, which is a property of the data, not of the graph."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from neurondb_amd import HnswIndex
    from neurondb_amd._lib import check, lib
    from oracle import ndbo
    n, dim, k = args.hnsw_nvec, args.dim, args.k

    def table(kind, rows, seed):
        x = make_data(rows, dim, kind, args.components, args.sigma, seed, 0x5EEDC0DE, dev)
        return x / x.norm(dim=1, keepdim=True)

    def recall_of(ix, base, q, efs, nr=1000, w16=False):    # (200 queries put the same graph anywhere in 0.88 .. 0.92: profiles/r03_h2_variants.txt)
        nr = min(nr, len(q))
        sims = q[:nr].double() @ base.double().T
        gt = torch.topk(sims, k, dim=1).indices.cpu().numpy() + 1
        out = {}
        for e in efs:
            ob, od, oc, oe = ix.search_intended(q[:max(nr, 256)], e, k, walk16=w16, strategy=strategy)
            out[e] = round(float(np.mean([len(set(ob[i, :oc[i]].tolist()) & set(gt[i].tolist())) / k for i in range(nr)])), 4)
        return out

    try:
        base = table("clustered", n, 0x5EED0003)
        q = table("clustered", nq, 0x5EED0004)
        r = np.random.default_rng(11).uniform(1e-12, 1.0, n)
        levels = np.clip((-np.log(r) * np.float32(0.36)).astype(np.int32), 0, 15)      # hnsw_am.c:1143-1161
        ix = HnswIndex(dim, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.build_intended(base, torch.arange(n, device=dev, dtype=torch.int64), levels, efc)
        tb = time.perf_counter() - t0
        sched = ix.build_stats()
        walk16 = dim % 4 == 0 and dim <= 1024       # the walk on fp16 walk rows (ndbhip_hnsw_search_intended_w16_device)
        reps = 3

        def timed(qq, w16):
            ix.search_intended(qq[:512], ef, k, walk16=w16, strategy=strategy)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = ix.search_intended(qq, ef, k, walk16=w16, strategy=strategy)
            return (time.perf_counter() - t0) / reps, r

        ts32, (pb, pd, pc, pe) = timed(q, False)
        if walk16:
            ts, (ob, od, oc, oe) = timed(q, True)
        else:
            ts, (ob, od, oc, oe) = ts32, (pb, pd, pc, pe)
        evals = float(oe.mean())
        rec = recall_of(ix, base, q, [ef], w16=walk16)[ef]
        rec32 = recall_of(ix, base, q, [ef])[ef] if walk16 else rec
        # steady state: four times the queries (a query is 400 .. 2500 evaluations and the device holds 3000 .. 4000 walks at a
        # time: a batch of 8192 is two or three rounds, as long as its unluckiest walker)
        big = {}
        try:
            qb = table("clustered", 4 * nq, 0x5EED0005)
            tb32, rb32 = timed(qb, False)
            big = {"queries": 4 * nq, "float4_walk_queries_per_s": round(4 * nq / tb32, 1)}
            if walk16:
                tb16, rb16 = timed(qb, True)
                big["queries_per_s"] = round(4 * nq / tb16, 1)
            del qb
        except Exception as e2:                      # noqa: BLE001
            big = {"error": f"{type(e2).__name__}: {e2}"}
        # two batches in flight: handles on ONE graph (ndbhip_hnsw_share), a host thread and a stream each — a batch ends with
        # its longest walks, the other lane's walks fill the device meanwhile
        flight = None
        try:
            import threading
            nl, rp = 2, 4
            handles = [ix, ix.share()]
            streams = [torch.cuda.Stream() for _ in range(nl)]
            errs, lastres = [], [None] * nl

            def lane(w):
                try:
                    check(lib().ndbhip_set_thread_stream(C.c_void_p(streams[w].cuda_stream)))
                    with torch.cuda.stream(streams[w]):
                        for _ in range(rp):
                            lastres[w] = handles[w].search_intended(q, ef, k, walk16=walk16, strategy=strategy)
                except Exception as ex:          # noqa: BLE001
                    errs.append(ex)
                finally:
                    lib().ndbhip_set_thread_stream(None)

            def go():
                th = [threading.Thread(target=lane, args=(w,)) for w in range(nl)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
                if errs:
                    raise errs[0]
            try:
                go()                                  # (the share's workspace grows here)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                go()
                torch.cuda.synchronize()
                tfl = (time.perf_counter() - t0) / (nl * rp)
                same = all(np.array_equal(r[0], ob) and np.array_equal(r[1].view(np.uint32), od.view(np.uint32)) and
                           np.array_equal(r[2], oc) for r in lastres)
                flight = {"batches_in_flight": nl, "queries_per_s": round(nq / tfl, 1), "ms_per_batch": round(tfl * 1e3, 3),
                          "identical_to_one_at_a_time": bool(same),
                          "how": "ndbhip_hnsw_share: handles on ONE graph, a host thread and a stream each"}
            finally:
                handles[1].close()
        except Exception as e4:                      # noqa: BLE001
            flight = {"error": f"{type(e4).__name__}: {e4}"}
        # oracle replay on the exported graph
        e = ix.export()
        vecs = np.zeros((n + 1, dim), np.float32)
        vecs[1:] = base.cpu().numpy()
        og = ndbo.HnswGraph.from_arrays(vecs, e["levels"], e["ncount"], e["nbrs"], None, e["entry_point"], e["entry_level"], m, efc)
        del vecs
        sample, bad, bad32 = 32, 0, 0
        qh = q[:max(sample, 1)].cpu().numpy()
        w16rows = og.walk_rows() if walk16 else None
        t0 = time.perf_counter()
        for i in range(sample):
            eb, ed, ns = og.search_intended_s(qh[i], strategy, ef, k)
            bad32 += not (pc[i] == len(eb) and np.array_equal(pb[i, :len(eb)], eb) and
                          np.array_equal(pd[i, :len(eb)].view(np.uint32), ed.view(np.uint32)) and pe[i] == ns)
        tc = (time.perf_counter() - t0) / sample
        if walk16:
            for i in range(sample):
                eb, ed, ns = og.search_intended_s(qh[i], strategy, ef, k, w16=w16rows)
                bad += not (oc[i] == len(eb) and np.array_equal(ob[i, :len(eb)], eb) and
                            np.array_equal(od[i, :len(eb)].view(np.uint32), ed.view(np.uint32)) and oe[i] == ns)
        else:
            bad = bad32
        del w16rows
        cores = host_cores()
        ncpu = int(min(nq, max(cores * 4, min(args.cpu_seconds, 10.0) * cores / max(tc, 1e-6)))) if args.cpu_seconds > 0 else 0
        cpu = None
        if ncpu > 0:
            qall = q[:ncpu].cpu().numpy()
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                list(ex.map(lambda lo: [og.search_intended_s(qall[i], strategy, ef, k) for i in range(lo, min(ncpu, lo + 8))], range(0, ncpu, 8)))
            cpu = {"value": round(ncpu / (time.perf_counter() - t0), 1), "unit": "queries/s", "cores": cores, "kind": "port",
                   "sample": f"{ncpu} of the same queries through oracle/ndb_oracle_hnsw2.c ndbo_h2_search on the exported "
                             "graph, one thread per core"}
        del og, e
        ix.close()
        evals32 = float(pe.mean())
        lists_b = lambda ev: (ev / (2 * m)) * (4 * 2 * m + 2)                      # noqa: E731
        bytes32 = evals32 * (4 * dim) + lists_b(evals32)
        # (walk rows: 2 bytes an element for the walk's evaluations, 4 for the ef re-scored entries)
        bytes_q = ((evals - ef) * (2 * dim) + ef * (4 * dim) + lists_b(evals - ef)) if walk16 else bytes32
        out = {"workload": f"HNSW {n}x{dim} fp32 m={m} ef_construction={efc} ef_search={ef} k={k}, "
                           f"{ {1: 'L2 (<->)', 2: 'cosine (<=>)', 3: 'inner product (<#>)'}[strategy]}: strategy {strategy} of the operator class, "
                           f"unit-norm rows, {nq}-query batches (BASELINE config C3), intended mode; table: mixture of {args.components} "
                           f"Gaussians sigma={args.sigma}, normalised",
               "strategy": strategy,
               "search": ("walk on fp16 walk rows (the reference's float4_to_fp16 image of the rows), the ef result entries re-scored "
                          "on the float4 rows: ndbhip_hnsw_search_intended_w16_device == oracle ndbo_h2_search_w16; `float4_walk` = "
                          "the walk on the float4 rows (ndbhip_hnsw_search_intended_device, what rounds 3-4 reported)") if walk16
               else "walk on the float4 rows",
               "build_vectors_per_s": round(n / tb, 1), "build_s": round(tb, 2),
               "build_schedule": {"batches": int(sched.get("batches", 0)), "largest_batch": int(sched.get("max_batch", 0))},
               "queries_per_s": flight["queries_per_s"] if flight and "queries_per_s" in flight else round(nq / ts, 1),
               "ms_per_batch": flight["ms_per_batch"] if flight and "ms_per_batch" in flight else round(ts * 1e3, 3),
               "one_batch_at_a_time": {"queries_per_s": round(nq / ts, 1), "ms_per_batch": round(ts * 1e3, 3),
                                       "note": "what rounds 3-4 reported as queries_per_s (the roofline below is this batch's)"},
               "evaluations_per_query": round(evals, 1), "recall_at_10": rec,
               "float4_walk": {"queries_per_s": round(nq / ts32, 1), "ms_per_batch": round(ts32 * 1e3, 3),
                               "evaluations_per_query": round(evals32, 1), "recall_at_10": rec32,
                               "oracle_mismatches": int(bad32), "bytes_per_query": int(bytes32)},
               "in_flight": flight,
               "steady_state": big,
               "roofline": h2_roofline(n, dim, m, ef, nq, ts, bytes_q, "k_h2_search_w16" if walk16 else "k_h2_search"),
               "oracle_parity": {"queries": sample, "mismatches": int(bad),
                                 "checked": "blocks, float4 distance bits, evaluation counts (graph equality: "
                                            "tests/test_gpu_hnsw2.py)"},
               "cpu_oracle_ms_per_query_single_thread": round(tc * 1e3, 3), "cpu_baseline": cpu}
        # the same build with batches of up to 32768 members (the schedule is a parameter of the definition: members of a
        # batch do not see one another): fewer, fuller launches for the build, and a graph the walks get through faster
        try:
            ix3 = HnswIndex(dim, m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix3.build_intended(base, torch.arange(n, device=dev, dtype=torch.int64), levels, efc, batch_max=32768)
            tb3 = time.perf_counter() - t0
            sched3 = ix3.build_stats()
            ix3.search_intended(q[:512], ef, k, walk16=walk16, strategy=strategy)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ix3.search_intended(q, ef, k, walk16=walk16, strategy=strategy)
            ts3 = (time.perf_counter() - t0) / reps
            out["batch_max_32768"] = {"build_vectors_per_s": round(n / tb3, 1), "build_s": round(tb3, 2),
                                      "build_schedule": {"batches": int(sched3.get("batches", 0)), "largest_batch": int(sched3.get("max_batch", 0))},
                                      "queries_per_s": round(nq / ts3, 1), "ms_per_batch": round(ts3 * 1e3, 3),
                                      "recall_at_10": recall_of(ix3, base, q, [ef], w16=walk16)[ef],
                                      "note": "ndbhip_hnsw_build_intended_device(..., batch_div 16, batch_max 32768); the numbers above are "
                                              "batch_max 8192 (the schedule rounds 3-4 measured); graph parity with the oracle under any "
                                              "schedule: tests/test_gpu_hnsw2.py"}
            ix3.close()
        except Exception as e3:                      # noqa: BLE001
            out["batch_max_32768"] = {"error": f"{type(e3).__name__}: {e3}"}
        del base
        torch.cuda.empty_cache()
        # the i.i.d. table, small: what the data does to any graph walk
        n2 = min(100_000, n)
        b2 = table("gauss", n2, 0x5EED0003)
        q2 = table("gauss", 512, 0x5EED0004)
        ix2 = HnswIndex(dim, m)
        r2 = np.random.default_rng(12).uniform(1e-12, 1.0, n2)
        t0 = time.perf_counter()
        ix2.build_intended(b2, torch.arange(n2, device=dev, dtype=torch.int64),
                           np.clip((-np.log(r2) * np.float32(0.36)).astype(np.int32), 0, 15), efc)
        tb2 = time.perf_counter() - t0
        out["iid_gauss_unit"] = {"rows": n2, "build_vectors_per_s": round(n2 / tb2, 1),
                                 "recall_at_10_by_ef": recall_of(ix2, b2, q2, [ef, 4 * ef, 16 * ef]),
                                 "note": "i.i.d. N(0,1) rows, normalised (BASELINE.md section 2): distances concentrate in 768 "
                                         "dimensions (the 10th neighbour is a few per cent nearer than the median row)"}
        ix2.close()
        return out
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def h2_roofline(n, dim, m, ef, nq, ts, bytes_q, kernel="k_h2_search"):
    """k_h2_search (csrc/ndbhip_hnsw2.h): one wave per query walks the graph best-first; every step is a dependent
    fetch of ~2m rows, so the kernel is bound by HBM latency x the waves in flight, not by bandwidth.  `achieved` =
    algorithmic bytes (distance evaluations counted by the kernel x row bytes + neighbour lists) / batch time; `traffic`
    = HBM-side bytes per batch from the committed PMC pass (2 x FETCH_SIZE + WRITE_SIZE per query x queries)."""
    import glob
    tr = src = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            e = json.load(f)["kernels"].get(kernel, {}).get("clustered_unit")
        if e and (e["workload"]["nvec"], e["workload"]["dim"], e["workload"]["m"], e["workload"]["ef"]) == (n, dim, m, ef):
            tr, src = int(e["traffic_bytes_per_query"]) * nq, "committed PMC pass " + os.path.relpath(path, ROOT)
            break
    ach = nq / ts * bytes_q / 1e9
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": tr, "traffic_source": src,
            "traffic_frac": None if not tr else round(tr / ts / 1e9 / HBM_PEAK_GBPS, 4),
            "bytes_per_query": int(bytes_q),
            "note": "bytes = distance evaluations (counted per query by the kernel) x row bytes + the neighbour lists read; "
                    "a dependent walk: what limits it is fetch latency x queries in flight (one wave per query, 8192 waves "
                    "over 256 CUs), so the fraction of the bandwidth roof stays far below 1 by construction"}


def hnsw_pmc_traffic(n, dim, m, ef, nq, seconds):
    """HBM-side bytes of one k_hnsw_search_fast launch from the newest committed PMC pass of this workload
    (profiles/*_pmc_traffic.json, written by tools/pmc_traffic_json.py from tools/pmc_hnsw.sh), and the HBM fraction
    that traffic gives over this run's batch time."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            k = json.load(f)["kernels"].get("k_hnsw_search_fast", {}).get("gauss_unit")
        if k and k.get("queries_per_dispatch") == nq and (n, dim, m, ef) == (1_000_000, 768, 16, 64):
            t = int(k["traffic_bytes_per_launch"])
            return {"traffic": t, "traffic_source": "committed PMC pass " + os.path.relpath(path, ROOT),
                    "traffic_frac": round(t / seconds / 1e9 / HBM_PEAK_GBPS, 4)}
    return {"traffic": None, "traffic_source": None}


def pmc_traffic(args, world, kernel="k_ivf_scan", data=None, want_busy=False):
    """(HBM-side bytes per launch, source) of the dominant kernel from the newest committed PMC pass that holds
    this kernel for this workload (profiles/*_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction
    applied) — counters cannot be read from inside the process being timed, so this is a measurement of an earlier
    run of the same binary and workload, labelled as such; (None, None) when there is none."""
    import glob
    none = (None, None, None) if want_busy else (None, None)
    if world != 1:
        return none
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            k = json.load(f)["kernels"].get(kernel, {}).get(data or args.data)
        if not k:
            continue
        w = k["workload"]
        same = (w["nvec"], w["dim"], w["lists"], w["probes"], w["batch"]) == \
            (args.nvec, args.dim, args.lists, args.probes, args.batch) and args.k == 10 and \
            w.get("rows", "f32") == args.rows and w.get("strategy", "l2") == args.strategy
        if same:
            src = "committed PMC pass " + os.path.relpath(path, ROOT)
            if want_busy:
                return int(k["traffic_bytes_per_launch"]), src, k.get("mfma_busy")
            return int(k["traffic_bytes_per_launch"]), src
    return none


def trace(what):
    """Progress on stderr (the JSON line is the only thing on stdout): which leg a run that dies was in."""
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {what}\n")
    sys.stderr.flush()


def host_cores():
    """Cores this process may really use: the smaller of the affinity mask and the cgroup's CPU quota (a container can
    see 256 CPUs and be allowed 8); os.cpu_count() alone overstates the CPU baseline's denominator."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g2:
                q, p2 = int(f.read()), int(g2.read())
            if q > 0:
                n = max(1, min(n, q // p2))
        except Exception:
            pass
    return n


def sweep_roofline(args, st, nq, steps, ms_per_step, data_kind, world):
    """Roofline of k_s16c_sweep (csrc/ndbhip_screen16c.h), the dominant kernel of a screened L2 batch over float4
    rows, on the work it DOES — every number recomputable from library_stats and profiles/:
      hbm   bytes = the row-plane tiles the batch touches, each counted once (library_stats.plane_bytes, counted on the
            device per launch: 128 rows x dim x 2 B per tile) / the launch's HIP-event time, against 8 TB/s;
      mfma  flops = 2 x dim per (row, pair) element actually multiplied (library_stats.rows_swept: one fp16 product per
            element, tile padding not counted) / the same time, against the dense fp16 peak 2.5 PFLOP/s.
    `bound` is the larger of the two fractions.  `traffic` = HBM-side bytes per launch from the newest committed PMC
    pass of this kernel and workload (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), `mfma.busy_pmc` the matrix
    pipe's busy share from the same passes.  SURVEY 8d's algorithmic bytes (every probed row once per query) are kept
    as `algorithmic`: the batch reads each touched tile once for all its queries and excludes most (query, sublist)
    pairs before the sweep, so that figure is a multiple of what is read, not a fraction of a roof."""
    launches = max(1, st["scan_launches"])
    ms = st["scan_kernel_ms"] / launches
    if ms <= 0:
        return None
    dim = args.dim
    dimp = (dim + 63) // 64 * 64
    plane = st["plane_bytes"] / launches
    pairs = st["rows_swept"] / launches
    issued = 2.0 * dimp * pairs
    hbm = plane / (ms * 1e-3) / 1e9
    mf = issued / (ms * 1e-3) / 1e12
    hf, mfr = hbm / HBM_PEAK_GBPS, mf / FP16_MFMA_PEAK_TFLOPS
    dense = st.get("dense_sweeps", 0) > 0
    wave = st.get("wave_sweeps", 0) > 0
    tr, src, busy = pmc_traffic(args, world, "k_s16c_dense" if dense else ("k_s16c_wsweep" if wave else "k_s16c_sweep"), data_kind, want_busy=True)
    alg = st["bytes_scored"] / launches
    how = {"ip": "; inner product: b = |q - x|^2 + M^2 - |x|^2 on the L2 layout's planes, thresholds in b's domain",
           "cosine": "; cosine: |q^ - x^|^2 over normalised planes"}.get(args.strategy, "") + \
        ("; fp16 mirror rows" if args.rows == "f16" else "")
    r = {"bound": "hbm" if hf >= mfr else "mfma",
         "kernel": ("k_s16c_dense (the centred one-plane sweep's dense tile, 256 pairs x 256 rows: loader / prefetcher waves, "
                    "the matrix pipe screens its own accumulator blocks, queued records; csrc/ndbhip_screen16d.h)" if dense else
                    ("k_s16c_wsweep (centred one-plane sweep as wave-autonomous register streams: fragment-major fp16 planes of "
                     "row - centre, a 32-row block x 32 pairs per wave, operands by coalesced 16-byte-per-lane loads straight into "
                     "the registers v_mfma_f32_32x32x16_f16 reads, items from per-XCD queues, every element's bounds in a place of "
                     "its own; csrc/ndbhip_screen16w.h)" if wave else
                     "k_s16c_sweep (centred one-plane sweep: fp16 planes of row - centre and query - centre, "
                     "v_mfma_f32_32x32x16_f16, operands by LDS DMA)")) + how,
         "achieved": round(hbm if hf >= mfr else mf, 1), "peak": HBM_PEAK_GBPS if hf >= mfr else FP16_MFMA_PEAK_TFLOPS,
         "unit": "GB/s" if hf >= mfr else "TFLOP/s", "frac": round(max(hf, mfr), 4),
         "traffic": tr, "traffic_source": src, "avg_launch_ms": round(ms, 4), "launches": int(launches),
         "hbm": {"bytes_per_launch": int(plane), "achieved": round(hbm, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": round(hf, 4),
                 "step_frac": round(plane / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "traffic_over_bytes": None if not tr else round(tr / max(1.0, plane), 3),
                 "note": "bytes = row-plane tiles touched, each once (device-counted); step_frac = the same bytes over "
                         "ms_per_step (everything outside the sweep included)"},
         "mfma": {"flops_per_launch": int(issued), "achieved": round(mf, 1), "peak": FP16_MFMA_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": round(mfr, 4), "busy_pmc": busy,
                  "pairs_swept_frac": round(st["rows_swept"] / max(1, st["rows_scored"]), 4),
                  "note": "flops issued for the (row, pair) elements the bounds left (one product per element)"},
         "algorithmic": {"bytes_per_launch": int(alg), "times_the_bytes_read": round(alg / max(1.0, plane), 1),
                         "note": "SURVEY 8d: rows probed x row bytes per query, no reuse across queries — not a "
                                 "fraction of a roof: a tile is read once per batch and most pairs are excluded first"},
         "rows_rescored_per_query": round(st.get("rows_rescored", 0) / max(1, nq * steps), 1),
         "rows_emitted_per_query": round(st.get("rows_emitted", 0) / max(1, nq * steps), 1)}
    return r


def build_cpu_baseline(args, base, cent_h, iters):
    """CPU baseline of the build leg: the oracle's assignment rule (ndbo_ivf_assign = ivf_am.c:905-935, the same loop
    k-means runs per Lloyd iteration) on all host cores over a bounded sample of the rows; the build's CPU time is
    (iterations x sample + N) rows at that rate (centroid updates are negligible next to it)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ndbo
    cores = host_cores()
    L = ndbo.lib()
    nl, dim = cent_h.shape
    cent = np.ascontiguousarray(cent_h, np.float32)
    t0 = time.perf_counter()
    probe = np.ascontiguousarray(base[:8].cpu().numpy(), np.float32)
    for r in range(8):
        L.ndbo_ivf_assign(cent, None, nl, nl, dim, probe[r], None)
    per_row = (time.perf_counter() - t0) / 8
    nrows = int(min(len(base), max(cores * 4, min(args.cpu_seconds, 10.0) * cores / max(per_row, 1e-6))))
    rows = np.ascontiguousarray(base[:nrows].cpu().numpy(), np.float32)
    out = np.zeros(nrows, dtype=np.int32)
    # oracle/ndb_oracle_mt.c: one pthread per core, 256 rows per grab (the rows are read once: no spread copy needed)
    wall = float(L.ndbo_mt_ivf_assign_batch(rows.ctypes.data, nrows, dim, cent, nl, cores, out))
    rate = nrows / wall                                   # rows assigned per second, all cores
    total_rows = args.nvec + iters * min(10000, 100 * args.lists, args.nvec)
    return {"value": round(args.nvec / (total_rows / rate), 1), "unit": "vectors/s", "cores": cores, "kind": "port",
            "sample": f"{nrows} rows assigned by oracle/ndb_oracle.c ndbo_ivf_assign on {cores} pthreads "
                      f"({rate:.0f} rows/s); build = ({iters} Lloyd iterations x sample + N) rows at that rate"}


def run_cpu_baseline(args, cent_h, list_len, rows_h, tid_h, qs, out_t, out_d, out_c):
    """Times the CPU oracle (oracle/, kind 'port') on the host cores for a bounded sample of the same workload in
    both builds BASELINE.md asks for — gcc -O2 (the reference's default PGXS flags: no -march, no FMA)
    and gcc -O3 -march=native (rebuilt on this host) — and checks the GPU results of the sample against it
    (ids + float4 bits).  `value` is the faster of the two."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ndbo
    cores = host_cores()
    tid_h = np.ascontiguousarray(tid_h).view(ndbo.TID_DTYPE).reshape(-1)
    off = np.zeros(len(list_len) + 1, dtype=np.int64)
    off[1:] = np.cumsum(list_len)
    img = ndbo.IvfImage(cent_h, off, rows_h, tid_h)
    q_h = qs.cpu().numpy()
    ndbo.lib()                                    # build/load outside the timed region
    native_ok = True
    try:                                          # -march=native must be compiled where it runs
        subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "_build/libndboracle_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ndbo.lib(native=True)
    except Exception:
        native_ok = False
    variants, n0, bad = {}, 0, 0
    for native in ([False, True] if native_ok else [False]):
        t0 = time.perf_counter()
        img.search(q_h[0], 1, args.probes, args.k, 0, native=native)
        one = time.perf_counter() - t0
        budget = args.cpu_seconds / (2 if native_ok else 1)
        # a warm-up of one query per core (it also makes the row copy whose pages the workers touch first), then the
        # timed sample: the C driver (oracle/ndb_oracle_mt.c), one pthread per core, queries handed out one by one
        img.search_batch_mt(q_h[:min(len(q_h), cores)], 1, args.probes, args.k, 0, nthreads=cores, native=native)
        _, _, _, w1 = img.search_batch_mt(q_h[:min(len(q_h), cores)], 1, args.probes, args.k, 0, nthreads=cores,
                                          native=native)
        per_round = max(w1, 1e-4)               # all cores busy with one query each
        nsample = int(max(cores, min(len(q_h), budget / per_round * cores)))
        nsample = min(nsample, len(q_h))
        et_all, ed_all, ec_all, wall = img.search_batch_mt(q_h[:nsample], 1, args.probes, args.k, 0, nthreads=cores,
                                                           native=native)
        # parity of the GPU results on this build's sample (ids + float4 bits)
        gt = ndbo.tids_from_device_u64(out_t[:nsample].cpu().numpy())
        gd = out_d[:nsample].cpu().numpy()
        gc = out_c[:nsample].cpu().numpy()
        vbad = 0
        for i in range(nsample):
            n_i = int(ec_all[i])
            ok = gc[i] == n_i and np.array_equal(gt[i, :n_i], ndbo.tids_to_u64(et_all[i, :n_i])) and \
                np.array_equal(gd[i, :n_i].view(np.uint32), ed_all[i, :n_i].view(np.uint32))
            vbad += (not ok)
        variants["gcc -O3 -march=native" if native else "gcc -O2 (reference default flags)"] = {
            "queries_per_s": round(nsample / wall, 2), "single_thread_ms_per_query": round(one * 1e3, 1),
            "speedup_over_one_thread": round(nsample / wall * one, 1),
            "sample_queries": nsample, "gpu_mismatches": int(vbad)}
        if not native:
            n0, bad = nsample, vbad
    img.free_spread()
    best = max(variants.values(), key=lambda v: v["queries_per_s"])
    return {"value": best["queries_per_s"], "unit": "queries/s", "cores": cores, "cpus_visible": os.cpu_count(), "kind": "port",
            "sample": f"{n0} queries of the same workload through oracle/ndb_oracle.c (-ffp-contract=off) on one pthread per "
                      "core (oracle/ndb_oracle_mt.c; the rows in memory first touched by the workers, 2 MiB stripes "
                      "round-robin over their NUMA nodes); both builds in `variants`, `value` = the faster",
            "variants": variants,
            "gpu_parity_on_sample": {"queries": n0, "mismatches": int(bad)}}


if __name__ == "__main__":
    main()
