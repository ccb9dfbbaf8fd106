#!/bin/bash
# kernel statistics of the i.i.d. step (tools/dense_probe.py under rocprofv3 --kernel-trace --stats), then SQ counters of the dense sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks_iid
(cd /tmp && STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "" > /tmp/ks_iid.log 2>&1)
f=$(find /tmp/ks_iid -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 45 > gpurun_out/r04_iid_kernel_stats.txt
grep -v amdgpu.ids /tmp/ks_iid.log | tail -2
head -34 gpurun_out/r04_iid_kernel_stats.txt | cut -c1-150
for v in "" "screen16_debug=2"; do
rm -rf /tmp/pmc_iid
(cd /tmp && STEPS=2 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "$v" > /tmp/pmc_iid.log 2>&1)
f=$(find /tmp/pmc_iid -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 45 > "gpurun_out/r04_iid_pmc_sq_${v:-default}.txt"
grep "k_s16c_dense" "gpurun_out/r04_iid_pmc_sq_${v:-default}.txt" | cut -c1-40,60-160
done
