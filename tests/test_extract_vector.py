"""Datum -> float4[] staging (ivfExtractVectorData, ivf_am.c:117-218): product host code vs oracle. CPU only."""
import ctypes as C
import struct

import numpy as np
import pytest

from neurondb_amd import _lib
from oracle import ndbo


def vector_datum(v):
    v = np.asarray(v, np.float32)
    return struct.pack("<ihh", 8 + 4 * len(v), len(v), 0) + v.tobytes()


def halfvec_datum(h):
    h = np.asarray(h, np.uint16)
    return struct.pack("<ih", 6 + 2 * len(h), len(h)) + h.tobytes()


def sparsevec_datum(total, idx, val):
    idx = np.asarray(idx, np.int32)
    val = np.asarray(val, np.float32)
    return struct.pack("<iii", 12 + 8 * len(idx), total, len(idx)) + idx.tobytes() + val.tobytes()


def bit_datum(bits):
    nb = len(bits)
    by = bytearray((nb + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            by[i // 8] |= 1 << (7 - i % 8)
    return struct.pack("<ii", 8 + len(by), nb) + bytes(by)


def both(kind, datum, cap=4096):
    d = np.frombuffer(datum, np.uint8).copy()
    out = np.full(cap, np.nan, np.float32)
    dim = C.c_int(-1)
    rc = _lib.lib().ndbhip_extract_vector(kind, d.ctypes.data, len(d), out.ctypes.data, cap, C.byref(dim))
    exp = np.full(cap, np.nan, np.float32)
    edim = C.c_int(-1)
    erc = ndbo.lib().ndbo_extract_vector(kind, d, exp, C.byref(edim))
    return rc, dim.value, out, erc, edim.value, exp


def test_vector_halfvec_sparsevec_bit_match_oracle():
    rng = np.random.default_rng(0)
    v = rng.standard_normal(37).astype(np.float32)
    h = rng.integers(0, 65536, 200).astype(np.uint16)
    h[:6] = [0x0001, 0x0200, 0x03FF, 0x8001, 0x7C00, 0xFC00]       # subnormals (quirk Q20), infinities
    h = h[(h & 0x7C00) != 0x7C00] if False else h
    cases = [(0, vector_datum(v)), (1, halfvec_datum(h)),
             (2, sparsevec_datum(50, [3, 49, 7, 60, -1, 7], [1.5, -2.0, 3.0, 9.0, 9.0, 4.0])),
             (3, bit_datum([1, 0, 0, 1, 1, 1, 0, 1, 1, 0, 1]))]
    for kind, datum in cases:
        rc, dim, out, erc, edim, exp = both(kind, datum)
        assert rc == 0 and erc == 0 and dim == edim
        a, b = out[:dim].view(np.uint32), exp[:dim].view(np.uint32)
        assert np.array_equal(a, b), kind


def test_sparsevec_drops_out_of_range_and_last_duplicate_wins():
    rc, dim, out, *_ = both(2, sparsevec_datum(8, [2, 9, -3, 2], [1.0, 5.0, 6.0, 7.0]))
    assert rc == 0 and dim == 8
    assert out[2] == 7.0 and out[:8].sum() == 7.0


def test_errors_are_codes_not_crashes():
    L = _lib.lib()
    dim = C.c_int(0)
    d = np.zeros(4, np.uint8)
    assert L.ndbhip_extract_vector(0, d.ctypes.data, 4, None, 0, C.byref(dim)) == _lib.ERR_INVALID
    assert L.ndbhip_extract_vector(9, d.ctypes.data, 4, None, 0, C.byref(dim)) == _lib.ERR_UNSUPPORTED
    v = np.frombuffer(vector_datum(np.ones(5)), np.uint8).copy()
    out = np.zeros(2, np.float32)
    assert L.ndbhip_extract_vector(0, v.ctypes.data, len(v), out.ctypes.data, 2, C.byref(dim)) == _lib.ERR_INVALID
    assert L.ndbhip_extract_vector(0, v.ctypes.data, len(v), None, 0, C.byref(dim)) == 0 and dim.value == 5
