#!/usr/bin/env python3
"""profiles/r1_pmc_traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE) of tools/run_prof_r1.sh:
HBM-side bytes per launch of the in-loop Chamfer kernel (steady state) and of the brute-force launch.
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE is reported in KB and counts half of the bytes of wide
coalesced reads -> x 1024 x 2; WRITE_SIZE in KB as reported -> x 1024."""
import json
import sqlite3
import sys


def per_launch(db, counter, sub, skip):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, sum(counter_value) from pmc_events where name like ? and counter_name = ? "
                     "group by dispatch_id order by dispatch_id", ("%" + sub + "%", counter)).fetchall()
    vals = [v for _, v in rows][skip:]
    return (sum(vals) / len(vals) if vals else None), len(vals)


def main(fetch_db, write_db, out, skip=30):
    skip = int(skip)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) of `python3 bench.py --steps 1 "
                     "--warmup 0 [--iters 60 in round 1] --no-cpu-baseline`; per-dispatch sums over XCDs, mean over the dispatches after the first "
                     f"{skip} (steady state) for the in-loop kernel, all dispatches for the brute-force launch",
           "correction": "gfx950: FETCH_SIZE x 1024 x 2 (KB; wide coalesced reads counted at half), WRITE_SIZE x 1024 (KB)"}
    for key, sub, sk in (("nn_in_loop", "nn_stream4", skip), ("nn_bruteforce", "nn_mfma_kernel", 0)):
        f, nf = per_launch(fetch_db, "FETCH_SIZE", sub, sk)
        w, nw = per_launch(write_db, "WRITE_SIZE", sub, sk)
        if f is None or w is None:
            continue
        res[key] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "dispatches": [nf, nw],
                    "bytes_per_launch": 2 * 1024 * f + 1024 * w}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
