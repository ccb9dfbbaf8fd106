#!/usr/bin/env python3
"""Per-kernel ISA facts from a -save-temps .s file: spills (v_readlane / v_writelane / scratch_), s_nop,
register counts.  usage: tools/isa_stats.py file.s [substring of the kernel symbol]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "k_ivf_scan_grouped"
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    sym = m.group(1)
    if want not in sym:
        continue
    end = s.index(".end_amdhsa_kernel", m.end())
    body = s[m.end():end]
    code = body[:body.index(".amdhsa_kernel")] if ".amdhsa_kernel" in body else body
    g = lambda k: (re.search(k + r"\s+(\d+)", body) or [None, "?"])[1]
    print(sym[:60], "lines", code.count("\n"), "readlane", code.count("v_readlane"), "writelane", code.count("v_writelane"),
          "scratch", code.count("scratch_"), "s_nop", code.count("s_nop"),
          "vgpr", g(r"\.amdhsa_next_free_vgpr"), "sgpr", g(r"\.amdhsa_next_free_sgpr"))
