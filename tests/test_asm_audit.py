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


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_dense_tile_for_full_tables_fits_the_register_file_without_scratch():
    """k_s16c_dense<0, 3, false> — the kernel `value` is measured on: 160 accumulator registers in the multiplier's copy of the
    item, 96 in the loader's — has to fit 256 VGPRs with NO private segment: a third copy of the item in the same kernel
    (round 6's first form of the small-tile map) spilled nine registers and cost 1.1 % on the i.i.d. table, which is why that
    map is a kernel of its own.  The small-tile kernel is allowed its few dwords."""
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "dev_asm.sh"), "k_s16c_dense<0,3,false>", "k_s16c_dense<0,3,true>"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    s = open(os.path.join(DEV, "dev.s")).read()
    bodies = {}
    for small in (0, 1):
        sym = f"_Z12k_s16c_denseILi0ELi3ELb{small}EEv"
        assert sym in s, sym
        body = s[s.index(".amdhsa_kernel " + sym):]
        bodies[small] = body[:body.index(".end_amdhsa_kernel")]
    assert ".amdhsa_private_segment_fixed_size 0" in bodies[0]
    vgpr = int(bodies[0].split(".amdhsa_next_free_vgpr")[1].split()[0])
    assert vgpr <= 256
    scratch = int(bodies[1].split(".amdhsa_private_segment_fixed_size")[1].split()[0])
    assert scratch <= 64
    # both run the two-product test instruction of pass 0 and the relative move that takes an element out
    fn = s[s.index("_Z12k_s16c_denseILi0ELi3ELb0EEv"):]
    fn = fn[:fn.index("s_endpgm")]
    assert fn.count("v_mfma_f32_32x32x2_f32") == 16 and "s_set_gpr_idx_on" in fn and "v_alignbit_b32" in fn
