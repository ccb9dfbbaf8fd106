import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.test_gpu_screen16w import clustered, image, index_of
from tests.util import oracle_image, oracle_search_batch, assert_same_results
from neurondb_amd import _lib
_lib.ensure_init(0)
L=_lib.lib(); check=_lib.check
for n,v in ((b"screen16c_wave", int(os.environ.get("WAVE","2"))),(b"screen16c_wave_min_nq",1),(b"screen16c_qb",1),(b"screen16_sub_min",300),(b"debug_s16",1)): check(L.ndbhip_set_option(n,v))
check(L.ndbhip_set_scan_mode(5))
dim,nprobe,strategy=128,6,2
rng=np.random.default_rng(500+dim+nprobe)
rows,lens=clustered(rng,dim); a=image(rows,lens)
q=(rows[rng.integers(0,len(rows),180)]+0.02*rng.standard_normal((180,dim))).astype(np.float32)
ix=index_of(a)
check(L.ndbhip_stats_reset())
t,d,c=ix.search(q,strategy,nprobe,10,0)
print({k:v for k,v in _lib.stats().items() if 'screen16' in k or 'wave' in k})
