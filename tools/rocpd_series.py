#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 rocpd .db, averaged over consecutive blocks of launches:
   tools/rocpd_series.py <db> <kernel substring> [block]"""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 50
c = sqlite3.connect(db)
rows = list(c.execute("select start, end from kernels where name like ? order by start", (f"%{pat}%",)))
d = [(e - s) / 1e3 for s, e in rows]
print(f"{pat}: {len(d)} dispatches, mean {sum(d)/max(len(d),1):.2f} us")
print("  per block of", blk, ":", " ".join(f"{sum(d[i:i+blk])/len(d[i:i+blk]):.1f}" for i in range(0, len(d), blk)))
