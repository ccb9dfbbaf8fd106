"""Pins the oracle's scalar distance recipes to the known-answer cases the
reference's own tests hold (tests/golden/known_answers.json lists the source
file:line of each).  CPU only."""
import json
import math
import os

import numpy as np
import pytest

from oracle import ndbo

with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
    CASES = json.load(f)["cases"]


def _eval_all(op, a, b):
    """Every recipe the reference has for this operator -> {name: value}."""
    L = ndbo.lib()
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    d = len(a)
    out = {}
    if op == "l2":
        out["ivf"] = L.ndbo_ivf_distance(a, b, d, 1)
        out["hnsw"] = L.ndbo_hnsw_distance(a, b, d, 1, None)
        for simd in (0, 8, 16):
            out[f"op{simd}"] = L.ndbo_op_l2(a, b, d, simd)
    elif op == "cosine":
        out["ivf"] = L.ndbo_ivf_distance(a, b, d, 2)
        out["hnsw"] = L.ndbo_hnsw_distance(a, b, d, 2, None)
        for simd in (0, 8, 16):
            out[f"op{simd}"] = L.ndbo_op_cosine(a, b, d, simd)
    elif op == "dot":            # vector_inner_product / `<#>` return +dot (Q15)
        for simd in (0, 8, 16):
            out[f"op{simd}"] = L.ndbo_op_ip(a, b, d, simd)
    elif op == "negdot":         # negative-inner-product convention
        out["hnsw"] = L.ndbo_hnsw_distance(a, b, d, 3, None)
        out["scalar"] = L.ndbo_op_ip_scalar(a, b, d)
        out["ivf_new"] = L.ndbo_ivf_distance(a, b, d, 3)
    return out


@pytest.mark.parametrize("case", CASES, ids=[f"{c['op']}-{i}" for i, c in enumerate(CASES)])
def test_known_answer(case):
    vals = _eval_all(case["op"], case["a"], case["b"])
    assert vals
    tol = case.get("tol", 0.0)
    exp = np.float32(case["expect"])
    for name, v in vals.items():
        if tol:
            assert abs(v - float(case["expect"])) <= tol, (name, v)
        else:
            # float4 result must be the correctly rounded expected value
            assert np.float32(v) == exp, (name, v, exp)
        if case["kind"].startswith("is:0"):
            # PostgreSQL prints float4 -0 as "-0": the reference test demands "0"
            assert not math.copysign(1.0, v) < 0 or case["op"] == "negdot", (name, v)


def test_cosine_zero_norm_conventions():
    L = ndbo.lib()
    z = np.zeros(4, np.float32)
    x = np.array([1, 2, 3, 4], np.float32)
    assert L.ndbo_ivf_distance(z, x, 4, 2) == 1.0          # ivf_am.c:1579-1580
    assert L.ndbo_hnsw_distance(z, x, 4, 2, None) == 2.0   # hnsw_am.c:1330-1331 (Q9)
    assert L.ndbo_op_cosine(z, x, 4, 0) == 1.0             # vector_distance.c:201-202


def test_fp16_roundtrip_and_subnormal_quirk():
    L = ndbo.lib()
    # normals decode like IEEE half
    for h in [0x3C00, 0xC000, 0x7BFF, 0x0400, 0x3555, 0x8000, 0x0000]:
        ref = np.array([h], np.uint16).view(np.float16)[0]
        assert np.float32(L.ndbo_fp16_to_float(h)) == np.float32(ref), hex(h)
    assert math.isinf(L.ndbo_fp16_to_float(0x7C00))
    # quirk Q20: subnormals come out 2^-10 too small (quantization.c:183-196)
    for h in [0x0001, 0x0200, 0x03FF]:
        ref = float(np.array([h], np.uint16).view(np.float16)[0])
        assert L.ndbo_fp16_to_float(h) == ref * 2.0 ** -10
    # encoder truncates and flushes (quantization.c:141-168)
    assert L.ndbo_float4_to_fp16(1.0) == 0x3C00
    assert L.ndbo_float4_to_fp16(1.0 + 2 ** -11 + 2 ** -12) == 0x3C00   # truncation, not RNE
    assert L.ndbo_float4_to_fp16(1e-6) == 0x0000                        # flush to zero
    assert L.ndbo_float4_to_fp16(1e6) == 0x7C00


def test_recipes_differ_in_rounding():
    """The three L2 recipes (Q8/Q9/Q16) are distinct roundings of the same value."""
    rng = np.random.default_rng(7)
    L = ndbo.lib()
    diff = 0
    for _ in range(200):
        a = rng.standard_normal(768).astype(np.float32)
        b = rng.standard_normal(768).astype(np.float32)
        v = [L.ndbo_ivf_distance(a, b, 768, 1), L.ndbo_hnsw_distance(a, b, 768, 1, None),
             L.ndbo_op_l2(a, b, 768, 0), L.ndbo_op_l2(a, b, 768, 8)]
        exact = math.sqrt(float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum()))
        for x in v:
            assert abs(x - exact) <= 2e-4 * exact
        # scalar Kahan path is the correctly rounded one
        assert np.float32(v[2]) == np.float32(exact)
        diff += len({np.float32(x).tobytes() for x in v}) > 1
    assert diff > 20


def test_ivf_distance_matches_numpy_sequential_fp32():
    """Independent restatement in numpy float32 scalars (sequential, unfused)."""
    rng = np.random.default_rng(11)
    L = ndbo.lib()
    for dim in (1, 3, 4, 28, 128, 257):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        s = np.float32(0)
        for i in range(dim):
            d = np.float32(a[i] - b[i])
            s = np.float32(s + np.float32(d * d))
        assert np.float32(L.ndbo_ivf_distance(a, b, dim, 1)) == np.sqrt(s, dtype=np.float32)
        assert np.float32(L.ndbo_ivf_l2sq(a, b, dim)) == s
        # hnsw: fp32 subtract widened, fp64 accumulate
        s64 = 0.0
        for i in range(dim):
            d = float(np.float32(a[i] - b[i]))
            s64 += d * d
        assert np.float32(L.ndbo_hnsw_distance(a, b, dim, 1, None)) == np.float32(math.sqrt(s64))


def test_pthread_driver_returns_what_single_calls_return():
    """oracle/ndb_oracle_mt.c adds no arithmetic: the batch on 4 threads (rows in the spread copy) = the loop."""
    import numpy as np
    from oracle import ndbo
    from tests.util import make_ivf_arrays, oracle_image
    a = make_ivf_arrays(3000, 48, 12, seed=5, dup_frac=0.1, integer=True)
    img = oracle_image(a)
    q = np.random.default_rng(6).integers(-3, 4, size=(37, 48)).astype(np.float32)
    t, d, c, wall = img.search_batch_mt(q, 1, 5, 10, 0, nthreads=4)
    assert wall > 0
    for i in range(len(q)):
        et, ed, _ = img.search(q[i], 1, 5, 10, 0)
        assert c[i] == len(et) and np.array_equal(ndbo.tids_to_u64(t[i, :c[i]]), ndbo.tids_to_u64(et))
        assert np.array_equal(d[i, :c[i]].view(np.uint32), ed.view(np.uint32))
    img.free_spread()
    cent = a["centroids"]
    out = np.zeros(len(a["rows"]), dtype=np.int32)
    ndbo.lib().ndbo_mt_ivf_assign_batch(a["rows"].ctypes.data, len(out), 48, cent, len(cent), 3, out)
    L = ndbo.lib()
    for r in range(0, len(out), 97):
        assert out[r] == L.ndbo_ivf_assign(cent, None, len(cent), len(cent), 48, a["rows"][r], None)
