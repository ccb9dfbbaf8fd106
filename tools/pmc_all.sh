#!/bin/bash
# The PMC passes of one state of the library (one counter group per run): tools/pmc_all.sh TAG [bench args]
# -> gpurun_out/pmc_TAG_{sq,sq2,tcc,fetch,write}.txt
tag=$1; shift
cd $GRAFT_REPO_ROOT
timeout 300 bash tools/pmc_pass.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@" > /dev/null 2>&1 </dev/null
timeout 300 bash tools/pmc_pass.sh ${tag}_sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "$@" > /dev/null 2>&1 </dev/null
timeout 300 bash tools/pmc_pass.sh ${tag}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@" > /dev/null 2>&1 </dev/null
timeout 300 bash tools/pmc_pass.sh ${tag}_fetch "FETCH_SIZE" "$@" > /dev/null 2>&1 </dev/null
timeout 300 bash tools/pmc_pass.sh ${tag}_write "WRITE_SIZE" "$@" > /dev/null 2>&1 </dev/null
for f in sq sq2 tcc fetch write; do echo "== $f"; grep -A60 "PMC counters" gpurun_out/pmc_${tag}_$f.txt | grep "k_s16_sweep\|k_s16_fin" | cut -c1-150; done
