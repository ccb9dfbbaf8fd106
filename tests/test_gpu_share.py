"""ndbhip_ivf_share (include/ndbhip.h): a second handle on the same mirror with scratch of its own — what two batches in
flight need (a thread, a stream, a handle each) without a second copy of the rows and planes.  Results through any handle
are the oracle's (ivfCollectCandidates, /root/reference/NeuronDB/src/index/ivf_am.c:1722-1909), concurrent batches on two
threads and streams equal the same batches one after the other, and while a share lives nothing persistent can change."""
import threading

import numpy as np
import pytest

from tests.test_gpu_screen16w import clustered, image, index_of
from tests.util import assert_same_results, oracle_image, oracle_search_batch

pytestmark = pytest.mark.gpu


@pytest.fixture
def lib():
    from neurondb_amd import _lib
    _lib.ensure_init()
    L = _lib.lib()
    _lib.check(L.ndbhip_set_option(b"screen16_sub_min", 300))
    yield _lib
    _lib.check(L.ndbhip_set_scan_mode(0))
    _lib.check(L.ndbhip_set_option(b"screen16_sub_min", 2048))
    _lib.check(L.ndbhip_set_option(b"screen16c_wave_min_nq", 1024))
    _lib.check(L.ndbhip_set_option(b"screen16c_qb", 0))
    _lib.check(L.ndbhip_set_thread_stream(None))


@pytest.mark.parametrize("rowtype,strategy", [("f32", 1), ("f32", 3), ("f16", 3), ("f32", 2)])
def test_a_share_answers_like_its_source_and_like_the_oracle(rowtype, strategy, lib):
    from oracle import ndbo
    rng = np.random.default_rng(5 + strategy)
    dim, nprobe, k = 128, 6, 10
    rows, lens = clustered(rng, dim)
    half = None
    if rowtype == "f16":
        half = rows.astype(np.float16).view(np.uint16)
        lut = np.array([ndbo.lib().ndbo_fp16_to_float(int(v)) for v in range(65536)], np.float32)
        rows = lut[half]
    a = image(rows, lens)
    q = (rows[rng.integers(0, len(rows), 200)] + 0.02 * rng.standard_normal((200, dim))).astype(np.float32)
    et, ed, ec, _ = oracle_search_batch(oracle_image(a), q, strategy, nprobe, k)
    ix = index_of(a, half)
    t, d, c = ix.search(q, strategy, nprobe, k)         # (lays the planes out, builds what this kind of batch needs)
    assert_same_results(t, d, c, et, ed, ec)
    sh = ix.share()
    lib.check(lib.lib().ndbhip_stats_reset())
    for h in (sh, ix, sh):
        t, d, c = h.search(q, strategy, nprobe, k)
        assert_same_results(t, d, c, et, ed, ec)
        t1, d1, c1 = h.search(q[:3], strategy, nprobe, k)        # (the exact path of a few queries)
        assert_same_results(t1, d1, c1, et[:3], ed[:3], ec[:3])
    st = lib.stats()
    # (nothing is laid out again; the cosine batches of this table fall back to the fp32 screen, share or not: its row norms
    # are the source's too)
    assert st["prepares"] == 0 and (strategy == 2 or (st["screen16_batches"] == 3 and st["screen16_fallbacks"] == 0)), str(st)
    sh.close()
    ix.close()


def test_two_threads_two_streams_one_mirror(lib):
    """batches in flight on a thread, a stream and a handle each — the source and its share — equal the same batches one
    after the other (and the oracle); the wave sweep and the ring both"""
    import torch
    rng = np.random.default_rng(77)
    dim, nprobe, k, nq, nb = 128, 7, 10, 192, 8
    rows, lens = clustered(rng, dim)
    a = image(rows, lens)
    qs = [(rows[rng.integers(0, len(rows), nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(np.float32) for _ in range(nb)]
    img = oracle_image(a)
    want = [oracle_search_batch(img, q, 1, nprobe, k)[:3] for q in qs]
    for wave in (0, 1):
        lib.check(lib.lib().ndbhip_set_option(b"screen16c_wave_min_nq", 1 if wave else 1024))
        lib.check(lib.lib().ndbhip_set_option(b"screen16c_qb", 1 if wave else 0))
        ix = index_of(a)
        ix.search(qs[0], 1, nprobe, k)
        handles = [ix, ix.share()]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got, err = [None] * nb, []

        def lane(w):
            try:
                lib.check(lib.lib().ndbhip_set_thread_stream(streams[w].cuda_stream))
                for rep in range(3):
                    for b in range(w, nb, 2):
                        got[b] = handles[w].search(qs[b], 1, nprobe, k)
                lib.check(lib.lib().ndbhip_set_thread_stream(None))
            except Exception as e:          # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=lane, args=(w,)) for w in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for b in range(nb):
            assert_same_results(*got[b], *want[b])
        handles[1].close()
        ix.close()


def test_a_shared_mirror_is_frozen(lib):
    from neurondb_amd._lib import NdbHipError
    from oracle import ndbo
    rng = np.random.default_rng(9)
    dim = 64
    rows, lens = clustered(rng, dim)
    a = image(rows, lens)
    q = (rows[rng.integers(0, len(rows), 64)] + 0.02 * rng.standard_normal((64, dim))).astype(np.float32)
    ix = index_of(a)
    # a share made before the planes exist cannot make them, and neither can its source while the share lives
    early = ix.share()
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    with pytest.raises(NdbHipError):
        early.search(q, 1, 6, 10)
    with pytest.raises(NdbHipError):
        ix.search(q, 1, 6, 10)
    early.close()
    t, d, c = ix.search(q, 1, 6, 10)
    lib.check(lib.lib().ndbhip_set_scan_mode(0))
    sh = ix.share()
    tid = ndbo.tids_from_rows(np.asarray([len(rows) + 1]))[0]
    for h in (ix, sh):
        with pytest.raises(NdbHipError):
            h.append(2, rows[5], tid)
        with pytest.raises(NdbHipError):
            h.delete(a["tids"][:3])
        with pytest.raises(NdbHipError):
            h.load(a["list_len"], a["rows"], a["tids"])
    with pytest.raises(NdbHipError):
        sh.share()                       # shares are made from the handle that owns the mirror
    # the library refuses to destroy a mirror with live shares (IvfIndex.close() therefore closes its shares first:
    # checked at the end, on a second share)
    assert lib.lib().ndbhip_ivf_destroy(ix._h) < 0 and b"shares" in lib.lib().ndbhip_last_error()
    # inner product was never run on the source: its constants are not there, and a frozen mirror cannot make them
    lib.check(lib.lib().ndbhip_set_scan_mode(5))
    with pytest.raises(NdbHipError):
        sh.search(q, 3, 6, 10)
    lib.check(lib.lib().ndbhip_set_scan_mode(0))
    t2, d2, c2 = sh.search(q, 1, 6, 10)
    assert t2.tobytes() == t.tobytes() and d2.tobytes() == d.tobytes() and np.array_equal(c2, c)
    sh.close()
    ix.append(2, rows[5], tid)           # thawed
    ix.search(q, 1, 6, 10)
    sh2 = ix.share()
    ix.close()                           # closes the share it made, then itself (the source may be collected first)
    assert sh2._h is None and ix._h is None
