#!/usr/bin/env python3
"""Per-kernel summary (count, total, avg, min, max, share) of a rocprofv3 rocpd
database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db).
Usage: tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.txt"""
import sqlite3
import sys


def main(path, top=30):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(workgroup_x) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} "
          f"{'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>14s} {'wg':>4s}")
    for r in rows[:top]:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} "
              f"{r[5] / 1e3:10.1f} {100 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} "
              f"{str(r[9]) + 'x' + str(r[10]):>14s} {r[11]:4d}")
    try:
        pmc = db.execute("select kernel_name, counter_name, sum(value), count(*), avg(value) from counters_collection "
                         "where kernel_name like 'k_%' or kernel_name like 'void k_%' "
                         "group by kernel_name, counter_name order by 1, 2").fetchall()
        if pmc:
            print("\n# PMC counters of this library's kernels: sum over dispatches, dispatches, avg per dispatch")
            print("# (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide")
            print("#  coalesced streams: double it before comparing with bytes - MI355X_MICROARCH.md, HBM section)")
            for r in pmc:
                print(f"{r[0][:60]:60s} {r[1]:24s} {r[2]:22.1f} {r[3]:6d} {r[4]:22.1f}")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
