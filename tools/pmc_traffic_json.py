#!/usr/bin/env python3
"""Turns the PMC summaries of one state of the library (tools/pmc_all.sh TAG -> gpurun_out/pmc_TAG_{fetch,write,tcc}.txt,
or copies of them under profiles/) into profiles/<round>_pmc_traffic.json, the file bench.py's roofline.traffic reads.

usage: tools/pmc_traffic_json.py PREFIX OUT.json STEPS [--hnsw PREFIX_HNSW]
  PREFIX       e.g. profiles/r02_pmc   (reads PREFIX_fetch.txt, PREFIX_write.txt, PREFIX_tcc.txt)
  STEPS        search steps the profiled bench.py ran in total (steps + warmup; tools/pmc_pass.sh: 3)

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are KiB per dispatch;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streams as 64 bytes, so read bytes =
2 x FETCH_SIZE x 1024.  The sweep runs twice a step (all queries, then the queries whose record buffer overflowed):
`traffic_bytes_per_launch` is the FIRST sweep's share, estimated as the step's traffic x (first sweep's time / both),
and `traffic_bytes_per_step` the sum of both."""
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


def main():
    prefix, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    wl = {"data": "clustered", "nvec": 1000000, "dim": 768, "lists": 1024, "probes": 32, "batch": 4096,
          "rows": "f32", "strategy": "l2"}
    doc = {"_comment": __doc__, "kernels": {}}
    e = entry(prefix, "k_s16_sweep<0, 0, 4, 2, 0, 0>", steps, wl, True)
    if e:
        doc["kernels"]["k_s16_sweep"] = {"clustered": e}
    e = entry(prefix, "k_s16_finalize", steps, wl, False)
    if e:
        doc["kernels"]["k_s16_finalize"] = {"clustered": e}
    if "--hnsw" in sys.argv:
        hp = sys.argv[sys.argv.index("--hnsw") + 1]
        nq = int(sys.argv[sys.argv.index("--hnsw") + 2])
        f = pick(counters(hp + "_fetch.txt"), "k_hnsw_search_fast")
        w = pick(counters(hp + "_write.txt"), "k_hnsw_search_fast")
        if f and w:
            fk, n = f["FETCH_SIZE"]
            wk, _ = w["WRITE_SIZE"]
            doc["kernels"]["k_hnsw_search_fast"] = {"gauss_unit": {
                "dispatches": n, "queries_per_dispatch": nq,
                "traffic_bytes_per_launch": int((2.0 * fk + wk) * 1024.0 / n),
                "traffic_bytes_per_query": int((2.0 * fk + wk) * 1024.0 / n / nq),
                "source": f"{hp}_{{fetch,write}}.txt"}}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
