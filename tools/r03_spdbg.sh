#!/bin/bash
# timing experiments on k_sub_pairs' count pass (screen16_debug 4 / 8 / 16: wrong results)
cd $GRAFT_REPO_ROOT
for v in 0 4 8 16 28; do
  PROF_LINES=60 bash tools/prof_pass.sh spd$v --build-from-host 0 --opt screen16_debug=$v 2>&1 | grep -E "k_sub_pairs<0>|k_pair_offsets" | cut -c1-130 | sed "s/^/dbg=$v /"
done
