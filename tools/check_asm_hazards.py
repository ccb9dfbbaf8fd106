#!/usr/bin/env python3
"""The query stream of the grouped kernels is fed by `asm volatile` scalar loads that the compiler does not model:
it believes the destination SGPRs are valid as soon as the asm statement has been issued.  If register pressure
makes it spill one of them (v_writelane) or read it (any use) before the matching `s_waitcnt lgkmcnt(0)`, the
kernel computes with garbage — silently.  The same goes for the asm global load of the cooperative bound pass
(VGPRs, `s_waitcnt vmcnt(0)`).  This checks the generated ISA: between an asm load and the next full wait on
its counter, no instruction may touch the registers it is filling.
usage: tools/check_asm_hazards.py file.s   (from `make -C neurondb_amd/csrc asm`)"""
import re
import sys


def regs(tok):
    """s[4:19] -> {('s',4)..('s',19)}, v7 -> {('v',7)}"""
    out = set()
    for kind, a, b, single in re.findall(r"\b([sv])\[(\d+):(\d+)\]|\b([sv]\d+)\b", tok):
        if kind:
            out.update((kind, i) for i in range(int(a), int(b) + 1))
        elif single:
            out.add((single[0], int(single[1:])))
    return out


def check(path, want=("k_ivf_scan_grouped", "k_ivf_bound_coop", "k_assign_grouped")):
    s = open(path).read()
    bad, seen = [], 0
    for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
        sym = m.group(1)
        if not any(w in sym for w in want):
            continue
        end = s.index(".end_amdhsa_kernel", m.end())
        body = s[m.end():end]
        code = body[:body.index(".amdhsa_kernel")] if ".amdhsa_kernel" in body else body
        pend_s, pend_v, in_asm = set(), set(), False
        for line in code.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART") or t.startswith(";APP"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND") or t.startswith(";NO_APP"):
                in_asm = False
                continue
            if re.match(r"^\.?LBB\w+:", t):      # a join point: what is pending on one path need not be on the other,
                pend_s.clear()                    # so the check is per basic block (the query stream's load and
                pend_v.clear()                    # its wait always share one)
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            ins = t.split(";")[0].strip()
            op = ins.split()[0] if ins else ""
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in ins:
                    pend_s.clear()
                if "vmcnt(0)" in ins:
                    pend_v.clear()
                continue
            if in_asm and op.startswith("s_load_dwordx"):
                pend_s |= regs(ins.split(",")[0])
                seen += 1
                continue
            if in_asm and op.startswith("global_load_dwordx"):
                pend_v |= regs(ins.split(",")[0])
                seen += 1
                continue
            if op.startswith("s_cbranch") or op == "s_branch":
                pend_s.clear()
                pend_v.clear()
                continue
            used = regs(ins)
            hit = used & (pend_s | pend_v)
            if hit and op not in ("s_nop",):
                bad.append((sym[:48], ins, sorted(hit)[:4]))
    return seen, bad


def check_m0(path, want=("k_s16_sweep",)):
    """k_s16_sweep writes M0 inside its LDS-DMA asm statements (s_mov_b32 m0 + global_load_lds_dwordx4) and hipcc
    does not model M0 as an asm clobber (it is a reserved register).  That is only safe while nothing the
    COMPILER emitted in the same kernel reads or writes M0: every mention of m0 must sit inside an asm block."""
    s = open(path).read()
    bad, seen = [], 0
    for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
        sym = m.group(1)
        if not any(w in sym for w in want):
            continue
        end = s.index(".end_amdhsa_kernel", m.end())
        body = s[m.end():end]
        code = body[:body.index(".amdhsa_kernel")] if ".amdhsa_kernel" in body else body
        in_asm = False
        for line in code.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART") or t.startswith(";APP"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND") or t.startswith(";NO_APP"):
                in_asm = False
                continue
            ins = t.split(";")[0]
            if re.search(r"\bm0\b", ins):
                if in_asm:
                    seen += 1
                else:
                    bad.append((sym[:48], ins.strip(), "m0 outside the DMA asm"))
    return seen, bad


def check_reserved(path, want=("k_s16c_wsweep",), first=88):
    """k_s16c_wsweep (csrc/ndbhip_screen16w.h) keeps its register stream in v88 .. v255, which only its own asm
    statements may name: the kernel is compiled with amdgpu_num_vgpr(44) (= v0 .. v87 for the compiler on gfx950).
    That is a promise of the register allocator, not of the language: every instruction OUTSIDE an asm block that
    names a register >= v88, and any scratch access at all (a spill would go through the in-order memory counter the
    stream's waits count on), is reported."""
    s = open(path).read()
    bad, seen = [], 0
    for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
        sym = m.group(1)
        if not any(w in sym for w in want):
            continue
        end = s.index(".end_amdhsa_kernel", m.end())
        body = s[m.end():end]
        code = body[:body.index(".amdhsa_kernel")] if ".amdhsa_kernel" in body else body
        in_asm = False
        for line in code.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART") or t.startswith(";APP"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND") or t.startswith(";NO_APP"):
                in_asm = False
                continue
            ins = t.split(";")[0].strip()
            if not ins or ins.startswith("."):
                continue
            if "scratch_" in ins:
                bad.append((sym[:48], ins, "scratch access"))
                continue
            high = [r for r in regs(ins) if r[0] == "v" and r[1] >= first]
            if in_asm:
                seen += 1 if high else 0
            elif high:
                bad.append((sym[:48], ins, "compiler code names the stream's registers"))
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(sys.argv[1])
    seen_m0, bad_m0 = check_m0(sys.argv[1])
    print(f"{seen_m0} M0 writes inside asm, {len(bad_m0)} uses of M0 by compiler-generated code")
    seen_r, bad_r = check_reserved(sys.argv[1])
    print(f"{seen_r} asm instructions on the register stream's own registers, {len(bad_r)} trespasses by compiler-generated code")
    bad = bad + bad_m0 + bad_r
    print(f"{seen} asm loads checked, {len(bad)} hazards")
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
