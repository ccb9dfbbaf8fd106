#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k dense_tile > gpurun_out/r04l_tests.log 2>&1
tail -3 gpurun_out/r04l_tests.log
timeout 900 python3 tools/dense_probe.py "" "screen16_debug=7" "screen16_debug=6" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04l_dense.txt
export TMPDIR=/tmp
rm -rf /tmp/ks_iid
(cd /tmp && STEPS=10 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_iid -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "" > /tmp/ks_iid.log 2>&1)
f=$(find /tmp/ks_iid -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 45 > gpurun_out/r04_iid_kernel_stats.txt
grep "qcprep\|seed_sample\|k_s16c_dense" gpurun_out/r04_iid_kernel_stats.txt | cut -c1-150
