"""The register-streaming sweep (csrc/ndbhip_screen16w.h) keeps operands in flight in registers that only its own
inline asm may name; the compiler is held to v0 .. v87 by amdgpu_num_vgpr.  That is the register allocator's
promise, so it is audited on the generated code: device-only compile of the kernel's instantiations, then
tools/check_asm_hazards.py (no compiler instruction on v88+, no scratch).  CPU only: hipcc cross-compiles."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "neurondb_amd", "csrc", "_dev")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_register_stream_registers_are_left_alone_by_the_compiler():
    kernels = [f"k_s16c_wsweep<{d},{ip},2>" for d in (2, 3, 4) for ip in ("false", "true")] + ["k_s16c_wsweep<2,false,3>", "k_s16c_wsweep<2,true,3>"]
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "dev_asm.sh")] + kernels, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), os.path.join(DEV, "dev.s")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    line = [l for l in out.stdout.splitlines() if "register stream" in l][0]
    assert int(line.split()[0]) > 1000 and " 0 trespasses" in line, out.stdout
    s = open(os.path.join(DEV, "dev.s")).read()
    # every instantiation present, none with a private segment
    for d, ip, blk in [(d, ip, 2) for d in (2, 3, 4) for ip in (0, 1)] + [(2, 0, 3), (2, 1, 3)]:
        sym = f"_Z13k_s16c_wsweepILi{d}ELb{ip}ELi{blk}EEv"
        assert sym in s
        body = s[s.index(sym):]
        body = body[body.index(".amdhsa_kernel"):body.index(".end_amdhsa_kernel")]
        assert ".amdhsa_private_segment_fixed_size 0" in body
        assert f".amdhsa_next_free_vgpr {256 if blk == 2 else 168}" in body
