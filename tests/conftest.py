import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def library_on_torch_stream():
    """The device-pointer entry points are asynchronous on the library's stream (include/ndbhip.h); the GPU tests
    fill and read their buffers with torch, so both run on torch's stream — the ordering contract a caller has
    to keep, and what bench.py does."""
    if not _have_gpu():
        yield
        return
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init(0)
    _lib.use_torch_stream()     # torch's default stream has handle 0 = "the library's own stream": see there
    for kv in filter(None, os.environ.get("NDBHIP_TEST_OPTS", "").split(",")):    # e.g. debug_build=1 (diagnostics only)
        name, _, val = kv.partition("=")
        _lib.check(_lib.lib().ndbhip_set_option(name.encode(), int(val or "1")))
    yield
    torch.cuda.synchronize()
    _lib.check(_lib.lib().ndbhip_set_stream(None))
