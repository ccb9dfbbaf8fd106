#!/bin/bash
# kernel statistics of the clustered headline step only (no gauss leg, no CPU baseline): tools/r03_kprof.sh TAG [bench args]
tag=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROF_LINES=70 bash tools/prof_pass.sh $tag --build-from-host 0 "$@" 2>&1 | cut -c1-125 | grep -v "at::native\|rocclr\|k_gen_rows\|k_assign_grouped\|k_s16_sweep<0, 0, 4, 2, 0, [12]>\|k_pack\|k_kmeans\|Cijk\|row_prep\|radius\|mid_\|k_cent_dups\|k_seq_sum\|assign_resolve" | head -36
