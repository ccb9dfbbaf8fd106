"""The C-ABI library loads without a GPU and exports everything include/ndbhip.h
declares; compute entry points refuse to run without a device (no CPU fallback)."""
import ctypes as C
import os

import numpy as np
import pytest

from neurondb_amd import _lib


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.lib_path()), "run __graft_entry__.build() first"
    # in-tree, not site-packages
    assert os.path.dirname(_lib.lib_path()).endswith(os.path.join("neurondb_amd", "lib"))


def test_exports_every_declared_symbol():
    L = C.CDLL(_lib.lib_path())
    syms = _lib.declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_binding_covers_header():
    L = _lib.lib()
    for s in _lib.declared_symbols():
        assert getattr(L, s).argtypes is not None, f"{s} has no ctypes signature"


def test_abi_version():
    assert _lib.lib().ndbhip_abi_version() == 1


def test_no_device_is_a_loud_error_not_a_fallback():
    L = _lib.lib()
    if L.ndbhip_device_count() > 0:
        pytest.skip("GPU present")
    assert L.ndbhip_init(0) == _lib.ERR_NODEVICE
    h = C.c_void_p()
    assert L.ndbhip_ivf_create(8, 4, C.byref(h)) == _lib.ERR_NODEVICE
    q = np.zeros(8, np.float32)
    out = np.zeros(1, np.float32)
    rc = L.ndbhip_batch_distance(q.ctypes.data, q.ctypes.data, out.ctypes.data, 1, 1, 8, 1, 0)
    assert rc == _lib.ERR_NODEVICE
    assert "init" in _lib.last_error()
    with pytest.raises(_lib.NdbHipError):
        _lib.check(rc)


def test_every_documented_option_is_accepted_and_bad_ones_are_refused():
    """ndbhip_set_option needs no device: the names include/ndbhip.h lists must be the names the library knows."""
    import ctypes as C
    import re
    from neurondb_amd import _lib
    L = _lib.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ndbhip.h")).read()
    end = hdr.index("int			ndbhip_set_option")
    block = hdr[hdr.rindex("\n/*", 0, end): end]          # the comment in front of the declaration
    names = set(re.findall(r'"([a-z0-9_]+)"', block))
    assert {"screen16", "screen16_records", "screen16_prune", "screen16_sublists", "build_screen16", "block_cache"} <= names
    defaults = {"screen16": 1, "screen16_records": 8192, "screen16_tighten": 1, "screen16_prune": 1, "screen16_sublists": 1,
                "screen16_sub_min": 256, "screen16_sub_rows": 128, "screen16_centered": 1, "cent_screen16": 1, "screen16_fin_threads": 64, "screen16_waves": 4,
                "probe_select_threads": 256, "probe_select_radix": 0, "build_screen16": 1, "block_cache": 1, "gchunk": 32,
                "scr_coop": 2, "scr_mfma": 1, "screen": 1, "screen16_slack": 1, "screen_min_nq": 5, "screen16_cosine": 1, "hnsw_intended_waves": 16, "screen16_cosine_centered": 1, "screen16_redo": 1,
                "screen16c_dense": 1, "screen16c_sample": 2048, "screen16c_tight": 128, "screen16c_epi": 1, "screen16_stage": 1, "screen16_ip_centered": 1,
                "screen16c_dense_min": 24, "screen16c_dense_min_sub": 100, "screen16c_wave": 2, "screen16c_wave_blocks": 2,
                "screen16c_wave_min_nq": 1024, "screen16c_plane_seeds": 1, "screen16_sweep_queue": 1, "screen16_sub_restrict": 0,
                "build_single_sweep": 1, "kmeans_screen16": 1, "slow_call_log": 0, "screen16c_dense_split": 3,
                "screen16c_dense_sync": 16, "screen16c_dense_small": 1, "screen16c_dense_spare": 0, "screen16c_bigk": 1, "scr_ch": 16}
    for n in sorted(names):
        rc = L.ndbhip_set_option(n.encode(), defaults.get(n, 0))
        assert rc == 0, (n, L.ndbhip_last_error().decode() if hasattr(L, "ndbhip_last_error") else rc)
    assert L.ndbhip_set_option(b"no_such_option", 1) != 0
    assert L.ndbhip_set_option(b"screen16_records", 3) != 0
    assert L.ndbhip_set_option(b"screen16_fin_threads", 100) != 0
    assert L.ndbhip_set_option(b"screen16_sub_rows", 1) != 0
    assert L.ndbhip_set_option(None, 1) != 0
