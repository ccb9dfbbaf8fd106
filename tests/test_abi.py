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
