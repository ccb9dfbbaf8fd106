#!/bin/bash
# Quick device-only compile of chosen kernel instantiations of ndbhip.hip's screened-scan headers, for reading the
# generated code and for tools/check_asm_hazards.py:   tools/dev_asm.sh 'k_s16c_wsweep<3,false>' 'k_s16c_sweep<1,3,0,1,0>' ...
# -> neurondb_amd/csrc/_dev/dev.s (the directory is scratch: git-ignored)
set -e
CS="$(cd "$(dirname "$0")/../neurondb_amd/csrc" && pwd)"
mkdir -p "$CS/_dev"
cd "$CS/_dev"
{
echo '#include <hip/hip_runtime.h>'
echo '#undef hipLaunchKernelGGL'
echo '#define hipLaunchKernelGGL(...) ((void)0)'
sed -n "1,$(grep -n 'include \"ndbhip_screen16w.h\"' ../ndbhip.hip | cut -d: -f1)p" ../ndbhip.hip | sed 's#"ndbhip_#"../ndbhip_#'
echo 'void *ndb_dev_keep[] = {'
for k in "$@"; do echo "(void *) $k,"; done
echo '};'
} > dev.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-unused-value -I.. -I../../../include --offload-device-only -S -o dev.s dev.hip 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|\^\|note:" | head -20 || true
