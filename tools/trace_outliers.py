#!/usr/bin/env python3
"""Per kernel of a rocprofv3 kernel trace (rocpd .db): calls, mean, median, max and WHERE the slowest launches sit -- a launch that costs
a hundred times its mean hides in any per-block average (round 5: the first search launch of a config-5 fit, 83 ms).
   tools/trace_outliers.py <db> [min calls]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(c.execute("select name, start, end from kernels order by start"))
by = {}
for n, s, e in rows:
    by.setdefault(n, []).append((e - s) / 1e3)
print(f"{'calls':>6} {'mean':>9} {'median':>9} {'max':>10} {'max/median':>10}  slowest launches (index: us)   kernel")
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < mn:
        continue
    sd = sorted(d)
    med = sd[len(sd) // 2]
    top = sorted(range(len(d)), key=lambda i: -d[i])[:3]
    print(f"{len(d):6d} {sum(d) / len(d):9.2f} {med:9.2f} {sd[-1]:10.2f} {sd[-1] / max(med, 1e-9):10.1f}  "
          + ", ".join(f"{i}: {d[i]:.0f}" for i in top) + f"   {n[:60]}")
