import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from neurondb_amd import IvfIndex, _lib
import tools.fuzz_scan as fz
_lib.ensure_init(0)
_lib.use_torch_stream()
rng = np.random.default_rng(12345)
last = {}
orig = IvfIndex.search
def wrapped(self, *a, **kw):
    last["ix"], last["a"], last["kw"] = self, a, kw
    return orig(self, *a, **kw)
IvfIndex.search = wrapped
n = 0
try:
    while n < 400:
        fz.one_case(rng, _lib.lib(), IvfIndex, _lib.check)
        n += 1
except AssertionError as e:
    print("failed at case", n, flush=True)
    L = _lib.lib()
    _lib.check(L.ndbhip_set_option(b"debug_s16", 1))
    _lib.check(L.ndbhip_stats_reset())
    t, d, c = orig(last["ix"], *last["a"], **last["kw"])
    print("rerun counts", c[:10], "stats", _lib.stats(), flush=True)
    _lib.check(L.ndbhip_set_option(b"screen16_centered", 0))
    t, d, c = orig(last["ix"], *last["a"], **last["kw"])
    print("uncentred counts", c[:10], flush=True)
