#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd .db (kernel trace / PMC) into the text summary committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    lines = [f"# rocprofv3 summary of {db}", ""]
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    lines.append("## kernel-trace --stats (durations in microseconds)")
    lines.append(f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'%':>7}  kernel")
    for name, calls, tot, avg, pct in cur:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        lines.append(f"{calls:7d} {tot:14.1f} {avg:12.2f} {pct:7.2f}  {short}")
    try:
        rows = list(c.execute(
            "select k.name, p.counter_name, count(*), avg(p.counter_value), sum(p.counter_value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name "
            "order by avg(p.counter_value) desc"))
    except Exception as e:  # schema differs between rocprofv3 builds
        rows = []
        try:
            cur = c.execute("select * from pmc_events limit 1")
            lines.append("pmc_events columns: " + ", ".join(d[0] for d in cur.description))
        except Exception:
            pass
    if rows:
        lines += ["", "## PMC counters (per-dispatch average, sum)"]
        for name, ctr, n, avg, tot in rows:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            lines.append(f"{ctr:>14} n={n:<6d} avg={avg:16.1f} sum={tot:18.1f}  {short}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
