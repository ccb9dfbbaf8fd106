"""The committed measurement artefacts are what bench.py's roofline.traffic reads: they must parse, agree with each other
and match the workload bench.py runs by default (no GPU needed)."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_traffic_json_is_what_the_summaries_say(tmp_path):
    out = tmp_path / "t.json"
    P = os.path.join(ROOT, "profiles")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_json.py"), str(out), "3",
                           "--ivf", "clustered", os.path.join(P, "r04_pmc_clustered"),
                           "--ivf", "gauss", os.path.join(P, "r04_pmc_gauss"),
                           "--c5", os.path.join(P, "r04_pmc_c5"),
                           "--h2", os.path.join(P, "r04_pmc_h2"), "8192"], stdout=subprocess.DEVNULL)
    fresh = json.load(open(out))["kernels"]
    kept = json.load(open(os.path.join(P, "r04_pmc_traffic.json")))["kernels"]
    for kern, sub in (("k_s16c_sweep", "clustered"), ("k_s16c_dense", "gauss"), ("k_s16_finalize", "clustered"),
                      ("k_s16c_sweep", "c5")):
        assert fresh[kern][sub]["traffic_bytes_per_launch"] == kept[kern][sub]["traffic_bytes_per_launch"]
    assert fresh["k_h2_search"]["clustered_unit"]["traffic_bytes_per_query"] == kept["k_h2_search"]["clustered_unit"]["traffic_bytes_per_query"]
    e = kept["k_s16c_sweep"]["clustered"]
    # 2 x FETCH_SIZE + WRITE_SIZE (KiB per step) is the per-step traffic the file reports
    assert abs((2 * e["fetch_kib_per_step"] + e["write_kib_per_step"]) * 1024 - e["traffic_bytes_per_step"]) < 4096
    assert 1e9 < e["traffic_bytes_per_launch"] < 4e9
    g = kept["k_s16c_dense"]["gauss"]
    # the matrix pipe's busy share is MFMA_BUSY / (1024 SIMDs x launch cycles); never above 1, and the dense table's
    # is the larger one (k_s16c_dense: the fp16 products and the fp32 screening instruction of its pass 0)
    assert 0.0 < e["mfma_busy"] < g["mfma_busy"] < 1.0
    assert 5e9 < g["traffic_bytes_per_launch"] < 3e10


def test_bench_finds_the_committed_traffic_for_its_default_workload():
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(data="clustered", nvec=1_000_000, dim=768, lists=1024, probes=32, batch=4096, k=10,
                                 rows="f32", strategy="l2")
    traffic, source = bench.pmc_traffic(args, 1, "k_s16c_sweep")
    assert traffic and "profiles/r04" in source
    t2, s2, busy = bench.pmc_traffic(args, 1, "k_s16c_dense", "gauss", want_busy=True)
    assert t2 > traffic and busy and "profiles/r06" in s2          # (round 6 re-profiled the dense tile: the newest pass wins)
    assert bench.pmc_traffic(args, 8, "k_s16c_sweep") == (None, None)          # a PMC pass describes one GPU
    args.strategy = "ip"
    assert bench.pmc_traffic(args, 1, "k_s16c_sweep") == (None, None)          # ... and one workload
    a5 = types.SimpleNamespace(data="clustered", nvec=10_000_000, dim=1536, lists=4096, probes=32, batch=256, k=10,
                               rows="f16", strategy="ip")
    t5, s5 = bench.pmc_traffic(a5, 1, "k_s16c_sweep", "c5")              # what bench.py's c5 leg looks up
    assert 1e9 < t5 < 4e9 and "profiles/r04" in s5
    h = bench.h2_roofline(1_000_000, 768, 16, 64, 8192, 16.3e-3, 2.6e6)
    assert h["traffic"] and h["traffic_source"] and 0 < h["traffic_frac"] < 1 and 0 < h["frac"] < 1


def test_the_committed_bench_line_is_one_json_object_with_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    r, c = d["roofline"], d["cpu_baseline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # every fraction of a roof in the line is a fraction
    assert 0 < r["frac"] <= 1 and 0 < r["hbm"]["frac"] <= 1 and 0 < r["mfma"]["frac"] <= 1 and 0 < r["hbm"]["step_frac"] <= r["hbm"]["frac"]
    g = d["iid_gauss"]
    assert 0 < g["roofline"]["frac"] <= 1 and g["cpu_baseline"]["value"] > 0 and g["oracle_parity"]["mismatches"] == 0
    assert c["gpu_parity_on_sample"]["mismatches"] == 0 and d["recall_at_10"] == 1.0
    assert "workload" in d["config"] and "model" not in d["config"]
    b = d["build"]
    assert b["searchable_vectors_per_s"] < b["vectors_per_s"] and b["lists_identical_to_exact_assignment"]


def test_round_5_artefacts_agree_with_each_other():
    """profiles/r05_*: the PMC traffic file is what the committed summaries say, the committed line carries the contract's
    fields plus the round's keys (serial, roofline.alone, c4, sigma_sweep, the HNSW walks), and bench.py finds the
    round's traffic for the kernel it runs now."""
    P = os.path.join(ROOT, "profiles")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.json")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_json.py"), out, "3",
                               "--ivf", "clustered", os.path.join(P, "r05_pmc_clustered"),
                               "--h2", os.path.join(P, "r05_pmc_h2"), "8192"], stdout=subprocess.DEVNULL)
        fresh = json.load(open(out))["kernels"]
    kept = json.load(open(os.path.join(P, "r05_pmc_traffic.json")))["kernels"]
    for kern in ("k_s16c_wsweep", "k_s16w_collect", "k_s16_finalize", "k_s16c_seed"):
        assert fresh[kern]["clustered"]["traffic_bytes_per_launch"] == kept[kern]["clustered"]["traffic_bytes_per_launch"]
    w = kept["k_s16c_wsweep"]["clustered"]
    assert 1.5e9 < w["traffic_bytes_per_launch"] < 2.0e9 and 0 < w["l2_hit_rate"] < 0.5
    f32, w16 = kept["k_h2_search"]["clustered_unit"], kept["k_h2_search_w16"]["clustered_unit"]
    assert fresh["k_h2_search_w16"]["clustered_unit"]["traffic_bytes_per_query"] == w16["traffic_bytes_per_query"]
    # the walk on fp16 rows moves a little more than half of the float4 walk's bytes (the re-score reads float4 rows)
    assert 0.5 < w16["traffic_bytes_per_query"] / f32["traffic_bytes_per_query"] < 0.7
    d = json.load(open(os.path.join(P, "r05_bench_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "serial", "c4", "c5", "sigma_sweep", "hnsw"):
        assert key in d, key
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= r["alone"]["frac"] <= 1
    # (the line read the PMC file committed when it ran; a later pass of the same kernel lands within a fraction of a per cent)
    assert abs(r["traffic"] - w["traffic_bytes_per_launch"]) < 0.01 * w["traffic_bytes_per_launch"] and 1.0 <= r["hbm"]["traffic_over_bytes"] <= 1.15
    assert d["serial"]["lanes_identical_to_serial"] and d["serial"]["ms_per_step"] > d["ms_per_step"]
    assert d["cpu_baseline"]["gpu_parity_on_sample"]["mismatches"] == 0 and d["recall_at_10"] == 1.0
    for name, leg in d["sigma_sweep"].items():
        if isinstance(leg, dict) and "queries_per_s" in leg:
            assert leg["recall_at_10"] >= 0.99 and leg["oracle_parity"]["mismatches"] == 0, name
    h = d["hnsw"]["intended"]
    assert h["oracle_parity"]["mismatches"] == 0 and h["float4_walk"]["oracle_mismatches"] == 0
    assert h["queries_per_s"] > h["float4_walk"]["queries_per_s"] and h["recall_at_10"] >= 0.90
    assert h["roofline"]["traffic_source"] and "r05" in h["roofline"]["traffic_source"]
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(data="clustered", nvec=1_000_000, dim=768, lists=1024, probes=32, batch=4096, k=10,
                                 rows="f32", strategy="l2")
    # (bench.py reads the NEWEST committed pass of the kernel and workload: round 6's now — test_round_6_artefacts_…)
    traffic, source = bench.pmc_traffic(args, 1, "k_s16c_wsweep")
    assert traffic and abs(traffic - w["traffic_bytes_per_launch"]) < 0.02 * traffic and "profiles/r0" in source



def test_round_6_artefacts_agree_with_each_other():
    """profiles/r06_*: the line bench.py printed on the GPU box (what the driver parses) is under 8 KB, strict JSON, carries the
    contract's fields with `roofline` and `cpu_baseline` as numbers, and is what driver_line() makes of the committed detail
    file; the headline is the i.i.d. table with the clustered table beside it; the PMC traffic file is
    what the committed summaries say and bench.py finds it for the kernel the headline runs."""
    P = os.path.join(ROOT, "profiles")
    raw = open(os.path.join(P, "r06_bench_line.json")).read().strip()
    assert "\n" not in raw and len(raw) < 8192

    def reject(tok):
        raise AssertionError(tok)
    d = json.loads(raw, parse_constant=reject)
    full = json.load(open(os.path.join(P, "r06_bench_detail.json")))
    sys.path.insert(0, ROOT)
    import bench
    assert bench.driver_line(full) == d
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "value_clustered", "roofline_clustered", "c4", "c5", "sigma_sweep", "hnsw"):
        assert key in d, key
    assert "i.i.d." in d["metric"] and d["config"]["data"].startswith("i.i.d.") and d["vs_baseline"] is None
    r, rc, c = d["roofline"], d["roofline_clustered"], d["cpu_baseline"]
    assert r["bound"] == "mfma" and r["kernel"] == "k_s16c_dense" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 0.7
    assert rc["bound"] == "hbm" and rc["kernel"] == "k_s16c_wsweep" and 0.45 < rc["frac"] < 0.8
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["gpu_mismatches_on_sample"] == 0
    assert d["value_clustered"] > 4 * d["value"] > 0 and d["recall_at_10"] > 0.85 and d["recall_at_10_clustered"] == 1.0
    assert d["serial"]["lanes_identical_to_serial"] and d["step_latency_ms"] >= d["ms_per_step"]
    for name, leg in d["sigma_sweep"].items():
        assert leg["recall_at_10"] >= 0.99 and leg["oracle_mismatches"] == 0, name
    # the middle of the sweep runs the dense tile now (round 6: from 48, later 24 pairs a bucket)
    assert d["sigma_sweep"]["s1.0"]["kernel"] == "k_s16c_dense" and d["sigma_sweep"]["s1.0"]["queries_per_s"] > 1.2e6
    h = d["hnsw"]
    assert h["strategy"] == 2 and h["recall_at_10"] >= 0.90 and h["intended_oracle_parity_mismatches"] == 0 and h["queries_per_s"] > 2e6
    assert h["ref_compat"]["oracle_mismatches"] == 0
    b = full["build"]
    assert b["lists_identical_to_exact_assignment"] and b["vectors_per_s"] > 3e7
    kept = json.load(open(os.path.join(P, "r06_pmc_traffic.json")))["kernels"]
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.json")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_json.py"), out, "3",
                               "--ivf", "clustered", os.path.join(P, "r06_pmc_clustered"), "--ivf", "gauss", os.path.join(P, "r06_pmc_gauss"),
                               "--h2", os.path.join(P, "r06_pmc_h2"), "8192"], stdout=subprocess.DEVNULL)
        fresh = json.load(open(out))["kernels"]
    g, w = kept["k_s16c_dense"]["gauss"], kept["k_s16c_wsweep"]["clustered"]
    assert fresh["k_s16c_dense"]["gauss"]["traffic_bytes_per_launch"] == g["traffic_bytes_per_launch"]
    assert fresh["k_s16c_wsweep"]["clustered"]["traffic_bytes_per_launch"] == w["traffic_bytes_per_launch"]
    assert 1.0e10 < g["traffic_bytes_per_launch"] < 2.5e10 and 0.4 < g["l2_hit_rate"] < 0.8 and 0.3 < g["mfma_busy"] < 0.8
    assert r["traffic"] == g["traffic_bytes_per_launch"] and rc["traffic"] == w["traffic_bytes_per_launch"]
    args = types.SimpleNamespace(data="gauss", nvec=1_000_000, dim=768, lists=1024, probes=32, batch=4096, k=10, rows="f32", strategy="l2")
    t, src, busy = bench.pmc_traffic(args, 1, "k_s16c_dense", "gauss", want_busy=True)
    assert t == g["traffic_bytes_per_launch"] and "profiles/r06" in src and busy == g["mfma_busy"]
    w16 = kept["k_h2_search_w16"]["clustered_unit"]
    assert 0.5 < w16["traffic_bytes_per_query"] / kept["k_h2_search"]["clustered_unit"]["traffic_bytes_per_query"] < 0.7
