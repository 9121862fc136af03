#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd .db): total, and the largest gaps with their neighbours.
   tools/trace_gaps.py <db> [min gap us to list]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = list(c.execute("select start, end, name from kernels order by start"))
busy = sum(e - s for s, e, _ in rows) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
gaps = [((rows[i + 1][0] - rows[i][1]) / 1e3, rows[i][2][:40], rows[i + 1][2][:40], i) for i in range(len(rows) - 1)]
print(f"{len(rows)} kernels, span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {sum(g[0] for g in gaps) / 1e3:.2f} ms")
small = [g[0] for g in gaps if g[0] < thr]
print(f"gaps below {thr} us: {len(small)}, mean {sum(small) / max(len(small), 1):.2f} us, total {sum(small) / 1e3:.2f} ms")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]:9.1f} us after #{g[3]} {g[1]}  -> {g[2]}")
