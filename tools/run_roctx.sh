#!/bin/bash
# roctx ranges under rocprofv3 (kernel + marker trace; no counters in this pass)
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_roctx
FDCAP_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --stats -d /tmp/prof_roctx -o r -- python3 /root/repo/bench.py --steps 1 --warmup 0 --value-only --iters 40 > $O/prof_roctx.log 2>&1
tail -3 $O/prof_roctx.log
python3 - <<'PY' > /root/repo/gpurun_out/r4_roctx_ranges.txt 2>&1
import sqlite3, glob, re
db = glob.glob('/tmp/prof_roctx/*results.db')[0]
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("# rocprofv3 --kernel-trace --marker-trace, FDCAP_ROCTX=1, bench.py --iters 40 (one fit)")
for t in names:
    if 'region' in t.lower() or 'marker' in t.lower():
        try:
            n = c.execute(f"select count(*) from {t}").fetchone()[0]
            print("table", t, n)
        except Exception as e:
            print("table", t, "?", e)
try:
    cols = [d[0] for d in c.execute("select * from regions limit 1").description]
    print("regions columns:", cols)
    for row in c.execute("select category, name, extdata from regions limit 3"):
        print("sample row:", row)
    import json, collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, ext, s0, e0 in c.execute("select name, extdata, start, end from regions"):
        label = name
        try:
            d = json.loads(ext) if ext else {}
            label = d.get("message") or d.get("msg") or name
        except Exception:
            pass
        if "fdcap" not in str(label) and ext and "fdcap" in str(ext):
            label = re.search(r"fdcap:[A-Za-z0-9_()]+", str(ext)).group(0)
        agg[label][0] += 1; agg[label][1] += (e0 - s0) / 1e3
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{n:6d} ranges  avg host span {tot / n:8.2f} us  {k}")
except Exception as e:
    print("regions query failed:", e)
PY
cat $O/r4_roctx_ranges.txt
