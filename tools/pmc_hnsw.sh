#!/bin/bash
# PMC passes of the HNSW search kernel at BASELINE config C3 size (one counter per run, --kernel-trace only):
# tools/pmc_hnsw.sh TAG  -> gpurun_out/pmc_TAG_hnsw_{fetch,write}.txt
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  rm -rf /tmp/pmch_$n
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmch_$n -o p -- python3 $GRAFT_REPO_ROOT/tools/hnsw_bench.py --nvec 1000000 --nq 8192 --oracle-sample 0 > /tmp/pmch_$n.log 2>&1 </dev/null
  f=$(find /tmp/pmch_$n -name "*.db" | head -1)
  if [ -z "$f" ]; then echo "no db for $n"; grep -v amdgpu /tmp/pmch_$n.log | tail -5; continue; fi
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_hnsw_$n.txt </dev/null
  grep "k_hnsw_search" $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_hnsw_$n.txt | cut -c1-160 | tail -4
done
