#!/bin/bash
# kernel statistics of single-query calls (tools/latency.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/latp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/latp -o p -- python3 $GRAFT_REPO_ROOT/tools/latency.py --n 300 > /tmp/latp.log 2>&1
f=$(find /tmp/latp -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f 60 | cut -c1-130 | grep -v "at::native\|k_gen_rows\|k_assign_grouped\|k_s16_sweep<0, 0, 4, 2, 0, [12]>\|k_pack\|k_kmeans\|Cijk\|row_prep\|radius\|mid_\|k_cent_dups\|k_seq_sum\|assign_resolve" | head -40
grep -v amdgpu /tmp/latp.log | tail -4
