#!/bin/bash
# Round 4's profiles in one box: kernel statistics and PMC passes of the clustered headline table, the i.i.d. N(0,1)
# table and the intended HNSW (each counter group in its own run, --kernel-trace only).
# usage: tools/r04_pmc.sh  -> gpurun_out/r04_*  (copy what is to be judged into profiles/)
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
  head -12 gpurun_out/${name}_kernel_stats.txt | cut -c1-140
}
B="$GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --build-from-host 0"
stats r04_bench $B
stats r04_iid $B --data gauss --steps 10 --warmup 2
stats r04_h2 $GRAFT_REPO_ROOT/tools/h2_bench.py 1000000 768 clustered 64
cp /tmp/ks_r04_h2.log gpurun_out/r04_h2_bench.log
KERNELS="k_s16c_sweep\|k_s16_fin\|k_s16c_dense" bash tools/pmc_all.sh r04c
KERNELS="k_s16c_sweep\|k_s16_fin\|k_s16c_dense" PASSES="sq tcc fetch write" bash tools/pmc_all.sh r04g --data gauss
KERNELS="k_h2_search" PASSES="fetch write" PROG="tools/h2_bench.py 1000000 768 clustered 64" bash tools/pmc_all.sh r04h
ls -la gpurun_out | grep "r04\|pmc_r04" | head -40
