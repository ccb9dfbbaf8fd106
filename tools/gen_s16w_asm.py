#!/usr/bin/env python3
"""Prints the per-slot inline-asm macros of neurondb_amd/csrc/ndbhip_screen16w.h (the block between the
"generated" markers): slot j of the register stream owns v[B + 32 j .. B + 32 j + 31] (B = 128: the S16W_ family,
four slots; B = 104: the S16W3_ family, two slots) — rows (operand B of
v_mfma_f32_32x32x16_f16) in the first 16, pairs (operand A) in the last 16, four registers per k-step.

  S16W_LDj(vo, rb, pv, qb)        the slot's eight requests (the stream's first D chunks)
  S16W_STj(acc, vo, rb, pv, qb, mk)   the slot's four matrix instructions on `acc`, its eight requests for the chunk
                                  that takes the slot next (the pairs' under the lane mask mk), and the wait states after
                                  which ordinary instructions may read the accumulators (the compiler does not know matrix
                                  instructions wrote them)"""
import sys

NT = "\\n\\t"


def row_loads(rb, a, b):
    return [f'"global_load_dwordx4 v[{rb + 4 * s}:{rb + 4 * s + 3}], %{a}, %{b}{" offset:%d" % (1024 * s) if s else ""}" S16W_RNT "{NT}"'
            for s in range(4)]


def pair_loads(pb, c, d):
    return [f'S16W_P("global_load_dwordx4 v[{pb + 4 * s}:{pb + 4 * s + 3}], %{c}, %{d}{" offset:%d" % (32 * s) if s else ""}{NT}")'
            for s in range(4)]


def gen(base=128, nslots=4, fam="S16W"):
    out = []
    for j in range(nslots):
        rb = base + 32 * j
        pb = rb + 16
        # (s_nop 4: a base that a v_readlane / v_readfirstlane has just written needs 5 wait states before a global_*
        # instruction reads it; the steps' requests follow four matrix instructions)
        ld = [f'"s_nop 4{NT}"'] + row_loads(rb, 0, 1) + pair_loads(pb, 2, 3) + ['"s_nop 0"']
        out.append(f"#define {fam}_LD{j}(vo, rb, pv, qb) asm volatile(" + " ".join(ld) +
                   ' :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")')
        st = [f'"s_nop 1{NT}"']
        for s in range(4):
            st.append(f'"v_mfma_f32_32x32x16_f16 %0, v[{pb + 4 * s}:{pb + 4 * s + 3}], v[{rb + 4 * s}:{rb + 4 * s + 3}], %0{NT}"')
        # the pairs' requests run under the mask of the lanes that hold a member (both k-halves): the address path's time
        # goes with the active lanes, and the other lanes' operand rows are never looked at (they keep what they had)
        st += row_loads(rb, 1, 2) + [f'"s_mov_b64 exec, %5{NT}"'] + pair_loads(pb, 3, 4) + [f'"s_mov_b64 exec, -1{NT}"']
        st.append(f'"s_nop 15{NT}s_nop 3"')
        out.append(f"#define {fam}_ST{j}(acc, vo, rb, pv, qb, mk) asm volatile(" + " ".join(st) +
                   ' : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")')
    return "\n".join(out)


if __name__ == "__main__":
    # two register maps: 2 blocks of 4 waves a compute unit (256 registers a lane: up to four slots from v128), and
    # 3 blocks (168 registers: two slots from v104)
    sys.stdout.write(gen(128, 4, "S16W") + "\n" + gen(104, 2, "S16W3") + "\n")
