#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" "screen16_debug=3" "screen16_debug=4" "screen16_debug=6"; do
rm -rf /tmp/pmc_iid
(cd /tmp && STEPS=2 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pmc_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "$v" > /tmp/pmc_iid.log 2>&1)
f=$(find /tmp/pmc_iid -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 45 > "gpurun_out/r04_iid_pmc_tcc_${v:-default}.txt"
echo "== ${v:-default}"; grep "k_s16c_dense" "gpurun_out/r04_iid_pmc_tcc_${v:-default}.txt" | cut -c1-30,60-160
rm -rf /tmp/pmc_iid
(cd /tmp && STEPS=2 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "$v" > /tmp/pmc_iid.log 2>&1)
f=$(find /tmp/pmc_iid -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 45 > "gpurun_out/r04_iid_pmc_fetch_${v:-default}.txt"
grep "k_s16c_dense" "gpurun_out/r04_iid_pmc_fetch_${v:-default}.txt" | grep FETCH | cut -c1-30,60-160
done
