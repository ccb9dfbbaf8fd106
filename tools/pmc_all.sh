#!/bin/bash
# The PMC passes of one state of the library (one counter group per run): tools/pmc_all.sh TAG [bench args]
# -> gpurun_out/pmc_TAG_{sq,sq2,tcc,fetch,write}.txt      (PASSES="sq fetch write" limits them)
tag=$1; shift
cd $GRAFT_REPO_ROOT
for p in ${PASSES:-sq sq2 tcc fetch write}; do
case $p in
sq) c="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE";;
sq2) c="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE";;
tcc) c="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";;
fetch) c="FETCH_SIZE";;
write) c="WRITE_SIZE";;
esac
timeout 400 bash tools/pmc_pass.sh ${tag}_$p "$c" "$@" > /dev/null 2>&1 </dev/null
echo "== $p"; grep -A80 "PMC counters" gpurun_out/pmc_${tag}_$p.txt | grep "${KERNELS:-k_s16c_sweep\|k_s16_fin}" | cut -c1-150
done
