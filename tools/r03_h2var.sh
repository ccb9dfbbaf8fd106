#!/bin/bash
# intended HNSW at 1M x 768 clustered: selection variants (build rate, q/s, recall), after the graph-equality tests
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_hnsw2.py -x -q -m gpu 2>&1 | tail -3
for v in "8192 5 16" "8192 7 16" "8192 1 16"; do
set -- $v
echo "== batch_max $1 select $2 batch_div $3"
H2_BMAX=$1 H2_SELECT=$2 H2_BDIV=$3 H2_NR=1000 timeout 900 python3 tools/h2_bench.py 1000000 768 clustered 64 96 2>&1 | grep -v amdgpu.ids | tail -3
done
