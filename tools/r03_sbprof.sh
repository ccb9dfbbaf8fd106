#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sb
NQS=256 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sb -o p -- python3 $GRAFT_REPO_ROOT/tools/small_batch_probe.py > /tmp/prof_sb.log 2>&1
grep "nq=" /tmp/prof_sb.log
f=$(find /tmp/prof_sb -name "*.db" | head -1)
python3 - <<PY
import sqlite3
db=sqlite3.connect("$f")
rows=db.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
tot=0
for r in rows:
    if r[1] in (16,32,48,64) or 'sweep' in r[0]:
        print(f"{r[0][:70]:70s} calls {r[1]:4d} avg_us {r[2]/1e3:8.1f} min {r[3]/1e3:8.1f}")
        tot+=r[2]*r[1]/16
print("sum per call (us)", tot/1e3)
PY
