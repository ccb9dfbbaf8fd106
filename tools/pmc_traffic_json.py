#!/usr/bin/env python3
"""Turns the PMC summaries of one state of the library (tools/pmc_all.sh TAG -> gpurun_out/pmc_TAG_{fetch,write,tcc}.txt,
or copies of them under profiles/) into profiles/<round>_pmc_traffic.json, the file bench.py's roofline.traffic reads.

usage: tools/pmc_traffic_json.py OUT.json STEPS [--ivf DATA PREFIX]... [--c5 PREFIX] [--h2 PREFIX NQ]
  PREFIX       e.g. profiles/r03_pmc_clustered   (reads PREFIX_fetch.txt, PREFIX_write.txt, PREFIX_tcc.txt, PREFIX_sq.txt)
  STEPS        search steps the profiled bench.py ran in total (steps + warmup; tools/pmc_pass.sh: 3)

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are KiB per dispatch;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streams as 64 bytes, so read bytes =
2 x FETCH_SIZE x 1024.  The centred sweep runs once a step (a second round only for queries whose record buffer
overflowed, none on the bench tables): `traffic_bytes_per_launch` = the pass's traffic / its dispatches."""
import json
import re
import sys


def counters(path):
    """{kernel prefix: {counter: (sum, dispatches)}} from a rocpd_summary.py file"""
    out, on = {}, False
    for line in open(path):
        if line.startswith("# PMC counters"):
            on = True
            continue
        if not on or line.startswith("#") or not line.strip():
            continue
        m = re.match(r"^(.*?)\s+([A-Z][A-Za-z0-9_]+)\s+([0-9.]+)\s+(\d+)\s+([0-9.]+)\s*$", line.rstrip())
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
    return out


def times(path):
    """{kernel prefix: (calls, total_ms, min_us, max_us)} from the kernel table at the top of the same file"""
    out = {}
    for line in open(path):
        if line.startswith("# PMC counters"):
            break
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+[0-9.]+\s+\d+\s+\d+\s+\d+", line.rstrip())
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)), float(m.group(6)))
    return out


def pick(table, needle):
    for k, v in table.items():
        if needle in k:
            return v
    return None


def entry(prefix, needle, steps, workload, two_rounds):
    f = pick(counters(prefix + "_fetch.txt"), needle)
    w = pick(counters(prefix + "_write.txt"), needle)
    t = pick(counters(prefix + "_tcc.txt"), needle)
    tm = pick(times(prefix + "_fetch.txt"), needle)
    if not f or not w:
        return None
    fetch_kib, nf = f["FETCH_SIZE"]
    write_kib, _ = w["WRITE_SIZE"]
    per_step = (2.0 * fetch_kib + write_kib) * 1024.0 / steps
    e = {"dispatches": nf, "steps": steps,
         "fetch_kib_per_step": round(fetch_kib / steps, 1), "write_kib_per_step": round(write_kib / steps, 1),
         "traffic_bytes_per_step": int(per_step), "workload": workload,
         "source": f"{prefix}_{{fetch,write,tcc}}.txt"}
    if two_rounds and tm and nf == 2 * steps:
        # dispatches alternate long (all queries) / short (overflowed queries); max_us ~ the long one
        calls, total_ms, _, max_us = tm
        share = min(1.0, max_us * steps / 1e3 / total_ms)
        e["first_sweep_time_share"] = round(share, 4)
        e["traffic_bytes_per_launch"] = int(per_step * share)
    else:
        e["traffic_bytes_per_launch"] = int(per_step * steps / nf)
    if t and "TCC_REQ_sum" in t and t["TCC_REQ_sum"][0] > 0:
        e["l2_hit_rate"] = round(t["TCC_HIT_sum"][0] / t["TCC_REQ_sum"][0], 4)
    return e


def busy(prefix, needle, nsimd=1024, nxcd=8):
    """share of the launch during which the matrix pipe of an average SIMD was busy: SQ_VALU_MFMA_BUSY_CYCLES (shader
    cycles, summed over the chip's 1024 SIMDs: 32 per v_mfma_f32_32x32x16_f16 — it equals 32 x SQ_INSTS_MFMA in the sq2
    pass) / (SIMDs x cycles of the launch).  GRBM_GUI_ACTIVE of the same pass is the launch's cycles summed over the 8
    XCDs (7.34 M for a 0.39 ms launch = 8 x 2.35 GHz x 0.39 ms), so cycles of the launch = GRBM_GUI_ACTIVE / 8."""
    try:
        c = pick(counters(prefix + "_sq.txt"), needle)
    except OSError:
        return None
    if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"][0] <= 0:
        return None
    return round(c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (nsimd * c["GRBM_GUI_ACTIVE"][0] / nxcd), 4)


def main():
    """usage: tools/pmc_traffic_json.py OUT.json STEPS [--ivf DATA PREFIX]... [--h2 PREFIX NQ]
       e.g.   tools/pmc_traffic_json.py profiles/r03_pmc_traffic.json 3 --ivf clustered profiles/r03_pmc_clustered \
                  --ivf gauss profiles/r03_pmc_gauss --h2 profiles/r03_pmc_h2 8192"""
    out, steps = sys.argv[1], int(sys.argv[2])
    doc = {"_comment": __doc__, "kernels": {}}
    a = sys.argv[3:]
    while a:
        if a[0] == "--ivf":
            data, prefix = a[1], a[2]
            a = a[3:]
            wl = {"data": data, "nvec": 1000000, "dim": 768, "lists": 1024, "probes": 32, "batch": 4096,
                  "rows": "f32", "strategy": "l2"}
            for needle, name in (("k_s16c_wsweep", "k_s16c_wsweep"), ("k_s16c_sweep", "k_s16c_sweep"), ("k_s16c_dense", "k_s16c_dense"),
                                 ("k_s16_finalize", "k_s16_finalize"), ("k_s16c_seed", "k_s16c_seed"), ("k_s16w_collect", "k_s16w_collect")):
                e = entry(prefix, needle, steps, wl, False)
                if e:
                    b = busy(prefix, needle)
                    if b is not None and name in ("k_s16c_sweep", "k_s16c_dense", "k_s16c_wsweep"):
                        e["mfma_busy"] = b
                    doc["kernels"].setdefault(name, {})[data] = e
        elif a[0] == "--c5":
            # BASELINE.md's C5 on one GPU (bench.py's c5 leg: tools/pmc_pass.sh with C5's arguments)
            prefix = a[1]
            a = a[2:]
            wl = {"data": "c5", "nvec": 10000000, "dim": 1536, "lists": 4096, "probes": 32, "batch": 256,
                  "rows": "f16", "strategy": "ip"}
            for needle, name in (("k_s16c_wsweep", "k_s16c_wsweep"), ("k_s16c_sweep", "k_s16c_sweep"), ("k_s16_finalize", "k_s16_finalize")):
                e = entry(prefix, needle, steps, wl, False)
                if e:
                    doc["kernels"].setdefault(name, {})["c5"] = e
        elif a[0] == "--h2":
            hp, nq = a[1], int(a[2])
            a = a[3:]
            cf, cw = counters(hp + "_fetch.txt"), counters(hp + "_write.txt")
            # (round 5: k_h2_search<0> walks the float4 rows, k_h2_search<NG> the fp16 walk rows; older files: one kernel)
            # (round 6: the kernels carry the strategy too — k_h2_search<0, 1>, k_h2_search<3, 1>)
            for needle, name in (("k_h2_search<0, 2>", "k_h2_search"), ("k_h2_search<3, 2>", "k_h2_search_w16"),
                                 ("k_h2_search<0,", "k_h2_search"), ("k_h2_search<3,", "k_h2_search_w16"), ("k_h2_search<0>", "k_h2_search"),
                                 ("k_h2_search<3>", "k_h2_search_w16"), ("k_h2_search(", "k_h2_search")):
                f, w = pick(cf, needle), pick(cw, needle)
                if f and w and name not in doc["kernels"]:
                    fk, n = f["FETCH_SIZE"]
                    wk, _ = w["WRITE_SIZE"]
                    # (tools/h2_bench.py: one 256-query warm-up launch + one of NQ per ef and walk; traffic is per query over both)
                    doc["kernels"][name] = {"clustered_unit": {
                        "dispatches": n, "queries_profiled": nq + 256,
                        "traffic_bytes_per_query": int((2.0 * fk + wk) * 1024.0 / (nq + 256)),
                        "workload": {"nvec": 1000000, "dim": 768, "m": 16, "ef": 64},
                        "source": f"{hp}_{{fetch,write}}.txt"}}
        else:
            raise SystemExit("unknown argument " + a[0])
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
