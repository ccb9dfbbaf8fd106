#!/bin/bash
# The artefacts under profiles/ for one round (run on the GPU box: gpurun -- 'bash tools/round_profiles.sh r05'):
# rocprofv3 kernel stats of the driver's command (steps in flight) and of the serial step, C4 / C5 kernel stats, PMC
# passes (each counter group in a run of its own, --kernel-trace only) over the sweep, collect, finalize and seeds, the
# per-wave trace of the wave sweep (profiling build, if neurondb_amd/lib_ph exists) and the overlap A/B.
# Copy what should be judged from gpurun_out/ to profiles/.
R=${1:-r06}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8     # (steps in flight need hardware queues of their own; rocprofv3's preloaded tool initialises the runtime before bench.py can set it)
stats() { tag=$1; shift; rm -rf /tmp/ks_$tag; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/ks_$tag.log 2>&1)
  f=$(find /tmp/ks_$tag -name "*.db" | head -1); [ -n "$f" ] && python3 tools/rocpd_summary.py $f 60 > gpurun_out/${R}_${tag}_kernel_stats.txt; grep -E "k_s16c_wsweep|k_s16w_collect|k_s16_finalize|k_s16c_seed" gpurun_out/${R}_${tag}_kernel_stats.txt | cut -c1-60,76-130; }
LEGS0="--hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --c4-nvec 0 --sigma-sweep 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0"
# (round 6: the driver's command times the i.i.d. table — BASELINE.md's — one step at a time; the clustered table is `--data clustered`)
echo "== kernel stats: the driver's command (i.i.d. N(0,1), k_s16c_dense)"; stats bench --gpus 1 --steps 20 --warmup 5 $LEGS0
grep -E "k_s16c_dense|k_s16c_qcprep|k_cent_select|k_s16_finalize" gpurun_out/${R}_bench_kernel_stats.txt | cut -c1-60,76-130
echo "== kernel stats: the clustered table, three steps in flight"; stats clustered --data clustered --gpus 1 --steps 20 --warmup 5 $LEGS0
echo "== kernel stats: the clustered table, one step after the other"; stats c2_serial --data clustered --gpus 1 --steps 20 --warmup 5 --inflight 1 $LEGS0
echo "== kernel stats: C4 on one GPU"; stats c4 --data clustered --nvec 10000000 --lists 4096 --steps 8 --warmup 3 --inflight 1 $LEGS0
echo "== kernel stats: C5 shape"; stats c5 --data clustered --nvec 10000000 --lists 4096 --dim 1536 --rows f16 --strategy ip --batch 256 --steps 20 --warmup 3 --inflight 1 $LEGS0
export KERNELS="k_s16c_dense\|k_s16_finalize\|k_s16c_qcprep"
PASSES="fetch write tcc sq" bash tools/pmc_all.sh ${R}_gauss --data gauss 2>&1 | tail -12
export KERNELS="k_s16c_wsweep\|k_s16w_collect\|k_s16_finalize\|k_s16c_seed"
PASSES="fetch write tcc sq" bash tools/pmc_all.sh ${R}_clustered --data clustered 2>&1 | tail -12
if [ -f neurondb_amd/lib_ph/libndbhip.so ]; then
  export NDBHIP_LIB=$GRAFT_REPO_ROOT/neurondb_amd/lib_ph/libndbhip.so
  for d in 2 3; do
    NDB_TRACE=gpurun_out/${R}_wtrace_d$d.npy timeout 600 python3 bench.py --data clustered --steps 20 --inflight 1 --opt screen16c_wave=$d $LEGS0 2>&1 >/dev/null | grep trace
  done
  unset NDBHIP_LIB
fi
python3 tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${R}_overlap.txt
# the intended HNSW (build + both walks at ef 64): kernel stats, FETCH_SIZE / WRITE_SIZE of the search kernels
rm -rf /tmp/ks_h2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_h2 -o p -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py 1000000 768 clustered 64 > /tmp/ks_h2.log 2>&1)
f=$(find /tmp/ks_h2 -name "*.db" | head -1); [ -n "$f" ] && python3 tools/rocpd_summary.py $f 30 > gpurun_out/${R}_h2_kernel_stats.txt; grep -E "k_h2_" gpurun_out/${R}_h2_kernel_stats.txt | cut -c1-60,76-130
export KERNELS="k_h2_search"
PROG="tools/h2_bench.py 1000000 768 clustered 64" PASSES="fetch write" bash tools/pmc_all.sh ${R}_h2 2>&1 | tail -8
