"""The centred sweep as wave-autonomous register streams (csrc/ndbhip_screen16w.h: k_s16c_wsweep, k_s16w_pairinfo,
k_s16w_collect) — what batches of a thousand queries and more take on clustered tables.  Forced onto the small batches
a test can replay on the oracle (screen16c_wave_min_nq = 1, screen16c_qb = 1), in every form it is built in (2, 3, 4
chunks in flight per wave; 1, 2, 3 blocks a compute unit), for the three operator classes, float4 and halfvec rows,
candidate caps, ties, NaN rows, deleted rows' holes and appended rows' spare blocks, member tiles with 1 to 32
and more than 32 pairs.  Results must be the oracle's bit for bit — ivfCollectCandidates,
/root/reference/NeuronDB/src/index/ivf_am.c:1722-1909 — and `wave_sweeps` must show that this kernel produced them."""
import numpy as np
import pytest

from tests.util import assert_same_results, oracle_image, oracle_search_batch

pytestmark = pytest.mark.gpu

FORMS = [(2, 2), (2, 3), (2, 1), (3, 2), (4, 2)]          # (chunks in flight, blocks a compute unit)


@pytest.fixture
def lib():
    from neurondb_amd import _lib
    _lib.ensure_init()
    L = _lib.lib()
    _lib.check(L.ndbhip_set_option(b"screen16c_wave_min_nq", 1))
    _lib.check(L.ndbhip_set_option(b"screen16c_qb", 1))
    _lib.check(L.ndbhip_set_option(b"screen16_sub_min", 300))
    _lib.check(L.ndbhip_set_scan_mode(5))
    yield _lib
    _lib.check(L.ndbhip_set_scan_mode(0))
    for name, val in ((b"screen16c_wave_min_nq", 1024), (b"screen16c_qb", 0), (b"screen16_sub_min", 2048), (b"screen16c_wave", 2),
                      (b"screen16c_wave_blocks", 2), (b"screen16_sublists", 1)):
        _lib.check(L.ndbhip_set_option(name, val))


def clustered(rng, dim, nlists=14, unstructured=3):
    """lists that mix 1 to 5 tight clusters each (regrouped into sublists from 300 rows up), one list without structure,
    duplicates inside lists (ties by position)"""
    comp = (rng.standard_normal((60, dim)) * 4).astype(np.float32)
    rows, lens = [], []
    for L in range(nlists):
        if L == unstructured:
            r = rng.standard_normal((900, dim)).astype(np.float32) * 4
        else:
            mine = rng.choice(60, 1 + L % 5, replace=False)
            n = 150 + 170 * len(mine)
            r = (comp[mine[rng.integers(0, len(mine), n)]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
            r[7] = r[3]
            r[n - 1] = r[3]
        rows.append(r)
        lens.append(len(r))
    return np.concatenate(rows), lens


def image(rows, lens):
    from oracle import ndbo
    nlists = len(lens)
    cents = np.stack([rows[sum(lens[:L]):sum(lens[:L + 1])].mean(0) for L in range(nlists)]).astype(np.float32)
    return dict(centroids=cents, list_len=np.asarray(lens, np.int64), rows=rows, tids=ndbo.tids_from_rows(np.arange(len(rows))))


def index_of(a, half=None):
    from neurondb_amd import IvfIndex
    ix = IvfIndex(a["centroids"].shape[1], len(a["list_len"]))
    ix.set_centroids(a["centroids"])
    if half is None:
        ix.load(a["list_len"], a["rows"], a["tids"])
    else:
        ix.load_f16(a["list_len"], half, a["tids"])
    return ix


def set_form(lib, wave, blocks):
    lib.check(lib.lib().ndbhip_set_option(b"screen16c_wave", wave))
    lib.check(lib.lib().ndbhip_set_option(b"screen16c_wave_blocks", blocks))


@pytest.mark.parametrize("wave,blocks", FORMS)
@pytest.mark.parametrize("strategy,cap,nprobe,dim,rowtype", [
    (1, 0, 6, 64, "f32"), (1, 40, 6, 128, "f32"), (3, 0, 4, 192, "f32"), (1, 0, 14, 768, "f32"), (1, 0, 6, 100, "f32"),
    (2, 0, 6, 128, "f32"), (3, 40, 6, 256, "f32"), (3, 0, 5, 128, "f16"), (1, 0, 6, 1536, "f16"), (2, 0, 5, 128, "f16sub")])
def test_wave_sweep_matches_the_oracle_in_every_form(wave, blocks, strategy, cap, nprobe, dim, rowtype, lib):
    rng = np.random.default_rng(500 + dim + nprobe)
    rows, lens = clustered(rng, dim)
    from oracle import ndbo
    half = None
    if rowtype != "f32":
        if rowtype == "f16sub":
            rows[::7, 3] = np.float32(3e-6)
        half = rows.astype(np.float16).view(np.uint16)
        lut = np.array([ndbo.lib().ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
        rows = lut[half]
    a = image(rows, lens)
    nq, k = 180, 10
    q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k, cap)
    set_form(lib, wave, blocks)
    ix = index_of(a, half)
    lib.check(lib.lib().ndbhip_stats_reset())
    t, d, c = ix.search(q, strategy, nprobe, k, cap)
    st = lib.stats()
    assert_same_results(t, d, c, et, ed, ec)
    # (cosine over rows that are near-copies of one another: more candidates inside the bound's error than a query's
    # survivor buffer holds, for the ring as for this kernel — the batch then goes to the older path, which is the design)
    assert st["screen16_batches"] + st["screen16_fallbacks"] == 1 and (strategy == 2 or st["screen16_fallbacks"] == 0), \
        str({kk: v for kk, v in st.items() if v})
    # (a chunk = 64 dimensions: rows of one chunk have nothing to keep in flight and take the ring; the cosine plane of a
    # halfvec mirror is the older two-plane sweep's)
    if (dim + 63) // 64 >= 2 and not (strategy == 2 and half is not None):
        assert st["wave_sweeps"] >= 1, st
    ix.close()


@pytest.mark.parametrize("wave,blocks", [(2, 2), (3, 2), (2, 3)])
def test_wave_sweep_with_holes_spare_blocks_a_nan_row_and_crowded_tiles(wave, blocks, lib):
    """Deleted rows leave holes in the planes, appended rows go into a bucket's spare blocks, a row that is not finite is
    always emitted; 200 queries drawn from TWO sublists give their tiles more than 32 members (several pair tiles per
    bucket), the rest a handful each."""
    from oracle import ndbo
    rng = np.random.default_rng(77)
    dim = 128
    rows, lens = clustered(rng, dim)
    a = image(rows, lens)
    off = np.concatenate([[0], np.cumsum(lens)])
    nq, k, nprobe = 260, 10, 7
    src = np.concatenate([off[5] + rng.integers(0, 40, 100), off[9] + rng.integers(0, 40, 100), rng.integers(0, len(rows), nq - 200)])
    q = (rows[src] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    set_form(lib, wave, blocks)
    ix = index_of(a)
    for strategy in (1, 3):
        et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k)
        lib.check(lib.lib().ndbhip_stats_reset())
        t, d, c = ix.search(q, strategy, nprobe, k)
        st = lib.stats()
        assert_same_results(t, d, c, et, ed, ec)
        assert st["wave_sweeps"] >= 1 and st["screen16_fallbacks"] == 0, st
    # appends (into spare blocks) and a delete (a hole), folded into the existing layout
    b = {k2: (v.copy() if hasattr(v, "copy") else v) for k2, v in a.items()}
    new_rows = (rows[off[5] + 3] + 0.001 * np.arange(1, 6)[:, None]).astype(np.float32)
    for i, r in enumerate(new_rows):
        tid = ndbo.tids_from_rows(np.asarray([len(rows) + i]))[0]
        ix.append(5, r, tid)
        end5 = off[6] + i
        b["rows"] = np.insert(b["rows"], end5, r, axis=0)
        b["tids"] = np.insert(b["tids"], end5, tid)
        b["list_len"][5] += 1
    dead = np.zeros(len(b["rows"]), bool)
    dead[off[9] + np.arange(2, 30, 3)] = True
    ix.delete(b["tids"][dead])
    keep = ~dead
    ll = b["list_len"].copy()
    o2 = np.concatenate([[0], np.cumsum(ll)])
    for L in range(len(ll)):
        ll[L] -= int(dead[o2[L]:o2[L + 1]].sum())
    b["rows"], b["tids"], b["list_len"] = b["rows"][keep], b["tids"][keep], ll
    et, ed, ec, _ = oracle_search_batch(oracle_image(b), q, 1, nprobe, k)
    lib.check(lib.lib().ndbhip_stats_reset())
    t, d, c = ix.search(q, 1, nprobe, k)
    st = lib.stats()
    assert_same_results(t, d, c, et, ed, ec)
    assert st["wave_sweeps"] >= 1 and st["screen16_fallbacks"] == 0 and st["prepares"] == 0, st
    ix.close()


def test_more_pairs_than_a_querys_list_holds_turns_the_mirror_to_the_ring(lib):
    """A query with more than S16_QP_CAP = 256 (query, sublist) pairs: the batch goes to the older path (results still the
    oracle's) and the mirror's later batches take the LDS ring — never a wrong answer, never an endless fallback."""
    rng = np.random.default_rng(91)
    dim, nlists = 64, 40
    # 40 lists of 8 far-apart clusters each, every query probes them all with list-level exclusion off: 320 pairs a query
    comp = (rng.standard_normal((nlists * 8, dim)) * 6).astype(np.float32)
    rows, lens = [], []
    for L in range(nlists):
        mine = np.arange(8 * L, 8 * L + 8)
        n = 8 * 60
        rows.append((comp[mine[np.arange(n) % 8]] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32))
        lens.append(n)
    rows = np.concatenate(rows)
    a = image(rows, lens)
    nq, k, nprobe = 150, 10, nlists
    q = (rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, 1, nprobe, k)
    lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", 0))
    try:
        ix = index_of(a)
        for _ in range(2):
            lib.check(lib.lib().ndbhip_stats_reset())
            t, d, c = ix.search(q, 1, nprobe, k)
            st = lib.stats()
            assert_same_results(t, d, c, et, ed, ec)
        assert st["wave_sweeps"] == 0 and st["screen16_batches"] == 1, st       # the second batch: the ring, no fallback
        ix.close()
    finally:
        lib.check(lib.lib().ndbhip_set_option(b"screen16_prune", 1))
