#!/usr/bin/env python3
"""Prints the per-slot inline-asm macros of neurondb_amd/csrc/ndbhip_screen16w.h (the block between the
"generated" markers): slot j of the register stream owns v[96 + 32 j .. 96 + 32 j + 31] — rows (operand B of
v_mfma_f32_32x32x16_f16) in the first 16, pairs (operand A) in the last 16, four registers per k-step."""
import sys


def gen(base=96, nslots=5):
    out = []
    nt = "\\n\\t"
    for j in range(nslots):
        rb = base + 32 * j
        pb = rb + 16
        ld = []
        for s in range(4):
            ld.append(f'"global_load_dwordx4 v[{rb + 4 * s}:{rb + 4 * s + 3}], %0, %1{" offset:%d" % (1024 * s) if s else ""}{nt}"')
            ld.append(f'"global_load_dwordx4 v[{pb + 4 * s}:{pb + 4 * s + 3}], %2, %3{" offset:%d" % (32 * s) if s else ""}{nt if s < 3 else ""}"')
        out.append(f"#define S16W_LD{j}(vo, rb, pv, qb) asm volatile(" + " ".join(ld) +
                   ' :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")')
        mm = [f'"s_nop 1{nt}"']
        for s in range(4):
            mm.append(f'"v_mfma_f32_32x32x16_f16 %0, v[{pb + 4 * s}:{pb + 4 * s + 3}], v[{rb + 4 * s}:{rb + 4 * s + 3}], %0{nt if s < 3 else ""}"')
        out.append(f"#define S16W_MM{j}(acc) asm volatile(" + " ".join(mm) + ' : "+v"(acc))')
    return "\n".join(out)


if __name__ == "__main__":
    sys.stdout.write(gen() + "\n")
