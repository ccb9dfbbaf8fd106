#!/bin/bash
# A/B of an environment switch on the same box: tools/ab_env.sh VAR A B [bench args]
v=$1; a=$2; b=$3; shift 3
for i in 1 2; do
  for x in $a $b; do
    env $v=$x python bench.py --steps 5 --warmup 2 --hnsw-nvec 0 --cpu-seconds 0 --recall-queries 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v=$x', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
