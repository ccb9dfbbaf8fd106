"""The grouped kernels feed their query stream (and the cooperative bound pass its row pieces) through
`asm volatile` loads the compiler does not model: it assumes the destination registers are valid once the asm
statement has been issued.  A build in which register pressure makes it touch one of them before the matching
s_waitcnt computes with garbage, silently — that is how a faster two-tile variant of the bound pass returned
wrong neighbours.  This test compiles the ISA (hipcc cross-compiles without a GPU) and
checks every asm load of the shipped kernels (tools/check_asm_hazards.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_instruction_touches_a_register_an_asm_load_is_filling():
    from tools.check_asm_hazards import check
    csrc = os.path.join(ROOT, "neurondb_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seen, bad = check(os.path.join(csrc, "_asm", "ndbhip-hip-amdgcn-amd-amdhsa-gfx950.s"))
    assert seen > 500, seen                      # the kernels with the hand-fed query stream are all there
    assert not bad, bad[:5]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_m0_belongs_to_the_lds_dma_asm_of_the_fp16_sweep():
    """k_s16_sweep sets M0 inside its LDS-DMA asm statements; hipcc does not model that (M0 is reserved), so the
    generated code of those kernels must not use M0 anywhere else."""
    from tools.check_asm_hazards import check_m0
    csrc = os.path.join(ROOT, "neurondb_amd", "csrc")
    asm = os.path.join(csrc, "_asm", "ndbhip-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(asm):
        subprocess.check_call(["make", "-C", csrc, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seen, bad = check_m0(asm)
    assert seen > 50, seen
    assert not bad, bad[:5]
