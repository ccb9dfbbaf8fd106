#!/usr/bin/env python3
"""How the kernels of steps in flight share the device: from a rocprofv3 --kernel-trace database of
`bench.py --inflight N` (kernels view: name, start, end, stream / queue), over the window that holds the LAST `nsweeps`
launches of the sweep kernel:
  - share of the window with at least one sweep running, with two or more, with any kernel running, with none;
  - per kernel name: launches, mean duration, mean number of OTHER kernels running beside it;
  - the gaps between consecutive sweeps (end of one -> start of the next) and what ran in them.
usage: tools/timeline_probe.py DB [SWEEP_NAME_PREFIX] [NSWEEPS]"""
import sqlite3
import sys
from collections import defaultdict


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def depth_profile(iv, lo, hi):
    """time spent at each concurrency depth inside [lo, hi]"""
    ev = []
    for s, e in iv:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            ev.append((s, 1))
            ev.append((e, -1))
    ev.sort()
    out, d, last = defaultdict(int), 0, lo
    for t, k in ev:
        out[d] += t - last
        last = t
        d += k
    out[d] += hi - last
    return out


def main(path, sweep="void k_s16c_wsweep", nsweeps=30):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    sw = [(s, e) for n, s, e in rows if n.startswith(sweep)]
    if len(sw) < 4:
        print("not enough sweeps", len(sw))
        return
    sw = sw[-nsweeps:]
    lo, hi = sw[0][0], sw[-1][1]
    win = hi - lo
    inside = [(n, s, e) for n, s, e in rows if e > lo and s < hi]
    allk = [(s, e) for n, s, e in inside]
    other = [(s, e) for n, s, e in inside if not n.startswith(sweep)]
    dp_s = depth_profile(sw, lo, hi)
    dp_a = depth_profile(allk, lo, hi)
    dp_o = depth_profile(other, lo, hi)
    print(f"window: {win / 1e3:.1f} us, {len(sw)} sweeps = {win / 1e3 / (len(sw) - 1):.1f} us from sweep to sweep")
    print("share of the window with k sweeps running: " + ", ".join(f"{k}: {v / win:.3f}" for k, v in sorted(dp_s.items())))
    print("share with k kernels of any kind running:   " + ", ".join(f"{k}: {v / win:.3f}" for k, v in sorted(dp_a.items()) if v / win >= 0.002))
    print("share with k kernels other than sweeps:     " + ", ".join(f"{k}: {v / win:.3f}" for k, v in sorted(dp_o.items()) if v / win >= 0.002))
    # per kernel: how many others run beside it (time-averaged)
    per = defaultdict(lambda: [0, 0, 0.0])
    for n, s, e in inside:
        ov = sum(max(0, min(e, e2) - max(s, s2)) for n2, s2, e2 in inside if (s2, e2, n2) != (s, e, n) and e2 > s and s2 < e)
        p = per[n[:60]]
        p[0] += 1
        p[1] += e - s
        p[2] += ov
    print(f"{'kernel':60s} {'n':>5s} {'avg_us':>8s} {'others beside it':>17s}")
    for n, (c, t, ov) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{n:60s} {c:5d} {t / c / 1e3:8.1f} {ov / max(1, t):17.2f}")
    gaps = [sw[i + 1][0] - sw[i][1] for i in range(len(sw) - 1)]
    gaps.sort()
    print(f"gaps between sweeps us: min {gaps[0] / 1e3:.1f} p50 {gaps[len(gaps) // 2] / 1e3:.1f} p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.1f} max {gaps[-1] / 1e3:.1f}; "
          f"sweep durations us: p50 {sorted(e - s for s, e in sw)[len(sw) // 2] / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "void k_s16c_wsweep", int(sys.argv[3]) if len(sys.argv) > 3 else 30)
