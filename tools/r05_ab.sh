#!/bin/bash
# round 5: A/B of two builds on one box, alternating:  tools/r05_ab.sh LIB_B [bench args]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
LIBB=$1; shift
run() { timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --steps 20 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'))" || tail -5 /tmp/err.txt; }
for i in 1 2 3; do
  unset NDBHIP_LIB; echo "A (lib)"; run "$@"
  export NDBHIP_LIB=$GRAFT_REPO_ROOT/neurondb_amd/$LIBB/libndbhip.so; echo "B ($LIBB)"; run "$@"
done
