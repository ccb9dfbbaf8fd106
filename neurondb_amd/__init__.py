"""neurondb_amd — MI355X (gfx950) vector-distance engine behind NeuronDB's
index access methods (IVFFlat list scan, HNSW traversal).

The compute path is the C-ABI library `lib/libndbhip.so` (hand-written HIP,
include/ndbhip.h).  This package is the thin host-side mirror used by the
tests, the bench and the multi-GPU driver; it never computes distances itself
and raises if the library is missing.
"""
from ._lib import NdbHipError, lib, lib_path, last_error  # noqa: F401
from .ivf import IvfIndex, IvfScan  # noqa: F401
from .hnsw import HnswIndex, HnswScan  # noqa: F401

__all__ = ["NdbHipError", "lib", "lib_path", "last_error", "IvfIndex", "IvfScan", "HnswIndex", "HnswScan"]
