import sys, os, pytest
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rc = pytest.main(["tests/test_gpu_screen16.py", "tests/test_gpu_ivf.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"])
from neurondb_amd import _lib
_lib.check(_lib.lib().ndbhip_set_option(b"debug_s16", 1))
rc = pytest.main(["tests/test_gpu_fullsize.py", "-x", "-q", "-m", "gpu", "-k", "iid_table", "-s", "-p", "no:cacheprovider"])
