#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "screen16_debug=6" "screen16_debug=3" "screen16_debug=2"; do
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
rm -rf /tmp/pmc_iid
(cd /tmp && STEPS=2 timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "$v" > /tmp/pmc_iid.log 2>&1)
f=$(find /tmp/pmc_iid -name "*.db" | head -1)
if [ -n "$f" ]; then python3 tools/rocpd_summary.py $f 45 > /tmp/o.txt; echo "== $v"; grep "k_s16c_dense" /tmp/o.txt | cut -c1-30,60-160; else grep -v amdgpu /tmp/pmc_iid.log | tail -3; fi
done
done 2>&1 | tee gpurun_out/r04_iid_pmc_tlb.txt
