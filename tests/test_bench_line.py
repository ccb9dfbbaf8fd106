"""The line bench.py prints is what the driver parses (an 8 KB tail of stdout: round 5's 36 KB line was UNMEASURED).
driver_line() is pure, so its size and strictness are checked here from committed detail files, without a device."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _reject(tok):
    raise AssertionError(f"non-JSON token {tok} in the line")


def _details():
    # r05's full line (the shape round 5 printed) and every bench_detail file committed since
    return [os.path.join(ROOT, "profiles", "r05_bench_line.json")] + sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_detail*.json")))


@pytest.mark.parametrize("path", _details(), ids=os.path.basename)
def test_the_printed_line_fits_the_driver_and_keeps_the_contract(path):
    import bench
    full = json.load(open(path))
    line = bench.driver_line(full)
    s = json.dumps(line, allow_nan=False)
    assert len(s) < bench.LINE_LIMIT < 8192, len(s)
    back = json.loads(s, parse_constant=_reject)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    r, c = back["roofline"], back["cpu_baseline"]
    assert r["frac"] == full["roofline"]["frac"] and r["bound"] in ("hbm", "mfma") and len(r["kernel"]) <= 80
    for key in ("achieved", "peak", "unit", "traffic", "avg_launch_ms"):
        assert key in r, key
    assert c["value"] == full["cpu_baseline"]["value"] and c["cores"] >= 1 and c["kind"] in ("port", "reference")
    assert "workload" in back["config"] and "model" not in back["config"]
    # one number per leg
    for leg in ("c4", "c5", "hnsw"):
        if isinstance(full.get(leg), dict) and "queries_per_s" in full[leg]:
            assert back[leg]["queries_per_s"] == full[leg]["queries_per_s"]
    if full.get("sigma_sweep"):
        assert len(back["sigma_sweep"]) == len(full["sigma_sweep"])
    assert not back.get("truncated")


def test_a_line_that_would_not_fit_drops_legs_not_the_contract():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    full["sigma_sweep"] = {f"sigma_{i}": dict(v, roofline=dict(v["roofline"], kernel="k" * 80))
                           for i in range(80) for v in [full["sigma_sweep"]["sigma_0.1"]]}
    line = bench.driver_line(full)
    s = json.dumps(line, allow_nan=False)
    assert len(s) < bench.LINE_LIMIT and line["truncated"] and "sigma_sweep" not in line
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def test_non_finite_numbers_never_reach_the_line(tmp_path, monkeypatch):
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    full["roofline"]["traffic"] = float("nan")
    full["c4"]["queries_per_s"] = float("inf")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, w = os.pipe()
    bench.emit(full, w)
    os.close(w)
    out = os.read(r, 1 << 16).decode()
    assert out.endswith("\n") and out.count("\n") == 1 and len(out) < bench.LINE_LIMIT
    back = json.loads(out, parse_constant=_reject)
    assert back["roofline"]["traffic"] is None and back["c4"].get("queries_per_s") is None
    json.loads(open(tmp_path / "bench_detail.json").read(), parse_constant=_reject)
