"""The reference's GPU plugin vtable (struct ndb_gpu_backend) filled from the device library
(include/ndb_backend.h): lifecycle, memory helpers and the launchers, called through the function pointers the
way src/gpu/common/gpu_distance.c and the k-means callers do.  Results must be the CPU fallbacks' bits."""
import ctypes as C

import numpy as np
import pytest

from oracle import ndbo

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_vtable_members_and_launchers():
    from neurondb_amd import _lib
    from neurondb_amd._lib import NdbHipDeviceInfo
    _lib.ensure_init()
    be = _lib.lib().ndb_hip_backend_get().contents
    assert be.name == b"ndbhip" and be.provider == b"AMD" and be.kind == 2      # NDB_GPU_BACKEND_ROCM
    assert not be.launch_quant_int8 and not be.launch_pq_encode                   # out of scope: NULL = CPU fallback
    assert be.is_available() == 1 and be.device_count() >= 1 and be.init() == 0
    info = NdbHipDeviceInfo()
    assert be.device_info(0, C.byref(info)) == 0 and info.compute_major == 9 and info.is_available and \
        info.total_memory_bytes > 2 ** 30
    assert be.device_info(99, C.byref(info)) < 0 and be.set_device(99) < 0 and be.set_device(0) == 0
    # memory helpers + streams
    p, s = C.c_void_p(), C.c_void_p()
    src = np.arange(1000, dtype=np.float32)
    dst = np.zeros_like(src)
    assert be.mem_alloc(C.byref(p), src.nbytes) == 0 and be.memcpy_h2d(p, _p(src), src.nbytes) == 0
    assert be.memcpy_d2h(_p(dst), p, src.nbytes) == 0 and np.array_equal(src, dst) and be.mem_free(p) == 0
    # the reference keeps its three stream members at the very end of ndb_gpu_backend: separate accessor
    sc, sy, sd = _lib.STREAM_CREATE(), _lib.STREAM_OP(), _lib.STREAM_OP()
    _lib.lib().ndb_hip_backend_streams(C.byref(sc), C.byref(sd), C.byref(sy))
    assert sc(C.byref(s)) == 0 and sy(s) == 0 and sd(s) == 0
    # launch_l2_distance / launch_cosine: n PAIRS, the CPU functions' arithmetic (Kahan L2, double cosine)
    O = ndbo.lib()
    rng = np.random.default_rng(3)
    for n, d in ((1, 768), (257, 33), (1000, 128)):
        A = (rng.standard_normal((n, d)) * rng.choice([1e-3, 1.0, 1e3], (n, 1))).astype(np.float32)
        B = rng.standard_normal((n, d)).astype(np.float32)
        B[0] = A[0]
        if n > 5:
            A[5] = 0.0
        out = np.zeros(n, np.float32)
        assert be.launch_l2_distance(_p(A), _p(B), _p(out), n, d, None) == 0
        exp = np.array([O.ndbo_op_l2(A[i], B[i], d, 0) for i in range(n)], np.float32)
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
        assert be.launch_cosine(_p(A), _p(B), _p(out), n, d, None) == 0
        exp = np.array([O.ndbo_op_cosine(A[i], B[i], d, 0) for i in range(n)], np.float32)
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
    assert be.launch_l2_distance(None, _p(B), _p(out), 1, 4, None) < 0
    assert be.launch_l2_distance(_p(A), _p(B), _p(out), 0, 4, None) < 0
    # launch_kmeans_assign / launch_kmeans_update: one Lloyd step == the oracle's kmeans_assign / update
    for n, d, k in ((3000, 64, 17), (500, 7, 5)):
        X = rng.standard_normal((n, d)).astype(np.float32)
        X[10] = X[3]
        Cn = X[:k].copy()
        Cn[k - 1] = 100.0                                     # an empty cluster: keeps its centroid
        idx = np.full(n, -1, np.int32)
        assert be.launch_kmeans_assign(_p(X), _p(Cn), _p(idx), n, d, k, None) == 0
        eidx = np.zeros(n, np.int32)
        ecnt = np.zeros(k, np.int32)
        O.ndbo_kmeans_assign(X, n, d, Cn, k, eidx, ecnt)
        assert np.array_equal(idx, eidx) and ecnt[k - 1] == 0
        C2 = Cn.copy()
        assert be.launch_kmeans_update(_p(X), _p(idx), _p(C2), n, d, k, None) == 0
        eC = Cn.copy()
        O.ndbo_kmeans_update(X, n, d, eidx, ecnt, k, eC)
        assert np.array_equal(C2.view(np.uint32), eC.view(np.uint32))
    # launch_quant_fp16: float4_to_fp16 (truncating, flush to zero), incl. edge values
    v = np.concatenate([rng.standard_normal(2000).astype(np.float32) * 10,
                        np.array([0.0, -0.0, 1e-6, -1e-6, 6.1e-5, 5.9e-5, 65504.0, 1e6, -1e6], np.float32)])
    h = np.zeros(len(v), np.uint16)
    assert be.launch_quant_fp16(_p(v), _p(h), len(v), None) == 0
    assert np.array_equal(h, np.array([O.ndbo_float4_to_fp16(float(x)) for x in v], np.uint16))
