import re,sys
s=open(sys.argv[1] if len(sys.argv)>1 else '_dev/dev.s').read()
want=sys.argv[2] if len(sys.argv)>2 else 'k_s16c'
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    sym=m.group(1)
    if want not in sym: continue
    end=s.index(".end_amdhsa_kernel", m.end())
    body=s[m.end():end]
    code = body[:body.index(".amdhsa_kernel")] if ".amdhsa_kernel" in body else body
    g=lambda k:(re.search(k+r"\s+(\d+)",body) or [None,"?"])[1]
    t=re.search(r"I(Li\d+E)+E",sym)
    print(sym[:28], t.group(0) if t else "", "lines",code.count("\n"),"scratch",code.count("scratch_"),"vgpr",g(r"\.amdhsa_next_free_vgpr"),"agpr_off",g(r"\.amdhsa_accum_offset"),"sgpr",g(r"\.amdhsa_next_free_sgpr"), "mfma16", code.count("v_mfma_f32_32x32x16"), "mfma_f32", code.count("v_mfma_f32_32x32x2"), "priv", g(r"\.amdhsa_private_segment_fixed_size"), "readlane", code.count("v_readlane"), "nop", code.count("s_nop"))
