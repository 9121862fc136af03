#!/usr/bin/env python3
"""Per-dispatch PMC totals of one kernel from a rocprofv3 rocpd .db: sums each counter over its
instances (XCD x SE), then averages over the dispatches from `skip` on (steady state of a run).
usage: pmc_kernel.py results.db kernel-substring [skip]"""
import sqlite3
import sys


def main(db, sub, skip=0):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                     "where name like ? group by dispatch_id, counter_name order by dispatch_id", ("%" + sub + "%",)).fetchall()
    ids = sorted({r[0] for r in rows})[int(skip):]
    keep = set(ids)
    agg, dur = {}, {}
    for d, n, v, t in rows:
        if d in keep:
            agg.setdefault(n, []).append(v)
            dur[d] = t
    print(f"{sub}: {len(ids)} dispatches (skipped {skip}), mean duration {sum(dur.values()) / max(len(dur), 1) / 1e3:.1f} us")
    for n, v in sorted(agg.items()):
        print(f"  {n:28s} {sum(v) / len(v):16.0f}")
    return agg


if __name__ == "__main__":
    main(*sys.argv[1:4])
