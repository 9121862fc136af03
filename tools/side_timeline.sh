#!/bin/bash
set -x
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_side
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_side -o r -- python3 /root/repo/bench.py --steps 1 --warmup 0 --value-only --profile-logging > /root/repo/gpurun_out/prof_side.log 2>&1
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/prof_side/*results.db')[0]
c = sqlite3.connect(db)
cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
print(cols)
rows = list(c.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
# find phase 2: last 900 kernels
t0 = rows[-700][1]
for name, s, e, q, st in rows[-700:-640]:
    short = name.replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:34]
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f}  q{q} s{st}  {short}")
PY
