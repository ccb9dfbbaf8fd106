#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab.sh LIB_A LIB_B [bench args]
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for l in $a $b; do
    NDBHIP_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 5 --warmup 2 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --recall-queries 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
