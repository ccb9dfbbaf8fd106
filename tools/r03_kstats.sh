#!/bin/bash
# kernel statistics only (the first half of tools/r03_pmc.sh): the clustered headline step, the i.i.d. table, single queries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
stats() { # NAME prog args...
  name=$1; shift
  rm -rf /tmp/ks_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o p -- python3 "$@" > /tmp/ks_$name.log 2>&1)
  f=$(find /tmp/ks_$name -name "*.db" | head -1)
  [ -n "$f" ] && python3 tools/rocpd_summary.py $f 40 > gpurun_out/${name}_kernel_stats.txt
  grep -v amdgpu.ids /tmp/ks_$name.log | tail -1 > gpurun_out/${name}_line.json
  head -16 gpurun_out/${name}_kernel_stats.txt | cut -c1-140
}
B="$GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --hnsw-nvec 0 --gauss-steps 0 --build-from-host 0"
stats r03_bench $B
stats r03_iid $B --data gauss --steps 10 --warmup 2
stats r03_single $GRAFT_REPO_ROOT/tools/latency.py --n 300
