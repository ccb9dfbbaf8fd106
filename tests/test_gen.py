"""The counter-based data generator (csrc/ndbhip_gen.h): pure function of (seed, row, dim), any slice in
any order, and the same bits from the host function (no device needed) and the device kernel."""
import ctypes as C

import numpy as np
import pytest


def _host(kind, seed, cseed, first, n, dim, comps=16, sigma=0.1):
    from neurondb_amd import _lib
    out = np.zeros((n, dim), np.float32)
    _lib.check(_lib.lib().ndbhip_gen_rows_host(kind, seed, cseed, first, n, dim, comps, sigma,
                                               out.ctypes.data_as(C.c_void_p)))
    return out


def test_host_generator_is_a_pure_function_of_its_counters():
    a = _host(0, 0x5EED0001, 0, 0, 300, 48)
    b = _host(0, 0x5EED0001, 0, 100, 50, 48)
    assert np.array_equal(a[100:150].view(np.uint32), b.view(np.uint32))          # any slice
    c = _host(0, 0x5EED0002, 0, 0, 300, 48)
    assert not np.array_equal(a, c)                                                # another seed, other data
    assert np.isfinite(a).all()


def test_host_generator_draws_standard_normals_and_the_mixture():
    x = _host(0, 12345, 0, 0, 4000, 64).astype(np.float64).ravel()
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1.0) < 0.01
    assert abs((np.abs(x) < 1.0).mean() - 0.6827) < 0.005 and abs((np.abs(x) > 3.0).mean() - 0.0027) < 0.001
    m = _host(1, 777, 999, 0, 3000, 32, comps=8, sigma=0.05)
    cen = _host(0, 999, 0, 0, 8, 32)                                              # the centers are z(center_seed, j * dim + d)
    d = ((m[:, None, :] - cen[None]) ** 2).sum(-1)
    assert (np.sqrt(d.min(1)) < 0.05 * np.sqrt(32) * 2).all()                     # every row sits at its center
    assert len(np.unique(d.argmin(1))) == 8


@pytest.mark.gpu
def test_device_generator_returns_the_hosts_bits():
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    for kind, first, n, dim in ((0, 0, 5000, 96), (1, 123456, 3000, 768), (0, 999_000, 1000, 33)):
        h = _host(kind, 0x5EED0001, 0x5EEDC0DE, first, n, dim, comps=1024, sigma=0.1)
        d = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().ndbhip_gen_rows_device(kind, 0x5EED0001, 0x5EEDC0DE, first, n, dim, 1024, 0.1,
                                                     C.c_void_p(d.data_ptr())))
        _lib.check(_lib.lib().ndbhip_synchronize())
        assert np.array_equal(d.cpu().numpy().view(np.uint32), h.view(np.uint32)), (kind, first, n, dim)
