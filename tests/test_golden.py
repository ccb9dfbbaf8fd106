"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them; GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from oracle import ndbo

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["l2", "cos", "ip", "refcompat", "k37"]


def _ivf_image(z):
    off = z["list_off"]
    order = np.argsort(z["assign"], kind="stable")
    return ndbo.IvfImage(z["centroids"], off, z["base"][order], ndbo.tids_from_rows(order)), order


def test_oracle_reproduces_ivf_golden():
    z = np.load(os.path.join(G, "ivf_small.npz"))
    img, asg, iters = ndbo.build_ivf_image(z["base"], int(z["nlists"]), max_iter=50)
    assert iters == int(z["kmeans_iters"])
    assert np.array_equal(img.centroids.view(np.uint32), z["centroids"].view(np.uint32))
    assert np.array_equal(asg, z["assign"])
    for name in CASES:
        s, nprobe, k, cap = (int(v) for v in z[f"{name}_params"])
        for i, q in enumerate(z["queries"]):
            t, d, _ = img.search(q, s, nprobe, k, cap)
            c = int(z[f"{name}_count"][i])
            assert len(t) == c
            assert np.array_equal(ndbo.tids_to_u64(t), z[f"{name}_tids"][i, :c])
            assert np.array_equal(d.view(np.uint32), z[f"{name}_dist"][i, :c].view(np.uint32))


def test_oracle_reproduces_hnsw_golden():
    z = np.load(os.path.join(G, "hnsw_small.npz"))
    g = ndbo.HnswGraph(z["vecs"].shape[1], m=int(z["m"]), ef_construction=int(z["efc"]), cap_nodes=len(z["vecs"]) + 2)
    for i, v in enumerate(z["vecs"]):
        g.insert(v, i, int(z["levels_in"][i]))
    a = g.arrays()
    assert np.array_equal(a["nbrs"], z["g_nbrs"]) and np.array_equal(a["ncount"], z["g_ncount"])
    assert a["entry_point"] == int(z["entry_point"]) and a["entry_level"] == int(z["entry_level"])
    for s in (1, 2, 3):
        for i, q in enumerate(z["queries"]):
            b, d, ns = g.search(q, s, 32, 10)
            c = int(z[f"s{s}_count"][i])
            assert np.array_equal(b, z[f"s{s}_blocks"][i, :c]) and ns == int(z[f"s{s}_scored"][i])
            assert np.array_equal(d.view(np.uint32), z[f"s{s}_dist"][i, :c].view(np.uint32))
    for ef, k in ((32, 10), (2, 5)):                       # src/scan/hnsw_scan.c: hnsw_search_layer
        for i, q in enumerate(z["queries"]):
            b, d, ns = g.search_layer(q, ef, k)
            c = int(z[f"layer{ef}_count"][i])
            assert np.array_equal(b, z[f"layer{ef}_blocks"][i, :c]) and ns == int(z[f"layer{ef}_scored"][i])
            assert np.array_equal(d.view(np.uint32), z[f"layer{ef}_dist"][i, :c].view(np.uint32))


@pytest.mark.gpu
def test_hip_ivf_reproduces_golden():
    from neurondb_amd import IvfIndex
    z = np.load(os.path.join(G, "ivf_small.npz"))
    n = len(z["base"])
    ix = IvfIndex(z["base"].shape[1], int(z["nlists"]))
    iters = ix.build(z["base"], ndbo.tids_from_rows(np.arange(n)))
    assert iters == int(z["kmeans_iters"])
    cent, ll, rows, tids = ix.export()
    assert np.array_equal(cent.view(np.uint32), z["centroids"].view(np.uint32))
    assert np.array_equal(ll, np.diff(z["list_off"]))
    for name in CASES:
        s, nprobe, k, cap = (int(v) for v in z[f"{name}_params"])
        t, d, c = ix.search(z["queries"], s, nprobe, k, cap)
        assert np.array_equal(c, z[f"{name}_count"])
        for i in range(len(c)):
            assert np.array_equal(ndbo.tids_to_u64(t[i, :c[i]]), z[f"{name}_tids"][i, :c[i]])
            assert np.array_equal(d[i, :c[i]].view(np.uint32), z[f"{name}_dist"][i, :c[i]].view(np.uint32))


@pytest.mark.gpu
def test_hip_hnsw_reproduces_golden():
    from neurondb_amd import HnswIndex
    z = np.load(os.path.join(G, "hnsw_small.npz"))
    nb = len(z["g_levels"])
    vecs = np.zeros((nb, z["vecs"].shape[1]), np.float32)
    vecs[1:] = z["vecs"]
    ix = HnswIndex(vecs.shape[1], int(z["m"]))
    ix.load(vecs, z["g_levels"], z["g_ncount"], z["g_nbrs"],
            z["g_tids"].astype(np.uint16).view(np.uint8).reshape(-1, 6), int(z["entry_point"]), int(z["entry_level"]))
    for s in (1, 2, 3):
        b, d, c, _, sc = ix.search(z["queries"], s, 32, 10)
        assert np.array_equal(c, z[f"s{s}_count"]) and np.array_equal(sc, z[f"s{s}_scored"])
        for i in range(len(c)):
            assert np.array_equal(b[i, :c[i]], z[f"s{s}_blocks"][i, :c[i]])
            assert np.array_equal(d[i, :c[i]].view(np.uint32), z[f"s{s}_dist"][i, :c[i]].view(np.uint32))


@pytest.mark.gpu
def test_hip_hnsw_search_layer_reproduces_golden():
    """hnsw_search_layer on a graph BUILT on the device from the golden inputs (a device-built mirror keeps the
    out-of-node slots the in-memory oracle graph has, which this search can read)."""
    from neurondb_amd import HnswIndex
    z = np.load(os.path.join(G, "hnsw_small.npz"))
    ix = HnswIndex(z["vecs"].shape[1], int(z["m"]))
    ix.build(z["vecs"], ndbo.tids_from_rows(np.arange(len(z["vecs"]))), z["levels_in"], int(z["efc"]))
    e = ix.export()
    assert np.array_equal(e["nbrs"], z["g_nbrs"]) and np.array_equal(e["ncount"], z["g_ncount"])
    for ef, k in ((32, 10), (2, 5)):
        b, d, c, _, sc = ix.search_layer(z["queries"], ef, k)
        assert np.array_equal(c, z[f"layer{ef}_count"]) and np.array_equal(sc, z[f"layer{ef}_scored"])
        for i in range(len(c)):
            assert np.array_equal(b[i, :c[i]], z[f"layer{ef}_blocks"][i, :c[i]])
            assert np.array_equal(d[i, :c[i]].view(np.uint32), z[f"layer{ef}_dist"][i, :c[i]].view(np.uint32))
