#!/usr/bin/env python3
"""profiles/r<N>_pmc_summary.json: per-launch PMC totals of the kernels bench.py's `roofline` / `blendshape_gemm` blocks describe,
from the rocprofv3 passes of tools/run_prof_r3.sh (each counter set its own pass, kernel-trace only):
  trace.db  --kernel-trace --stats            -> duration_us_trace (average over the run)
  fetch.db  --pmc FETCH_SIZE                  -> FETCH_SIZE_KB
  write.db  --pmc WRITE_SIZE                  -> WRITE_SIZE_KB
  sq.db     --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY
Counters are summed over their instances (XCD x SE) per dispatch and averaged over the dispatches after `skip` (in-loop NN
kernel: steady state of a fit) or over all of them.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section):
FETCH_SIZE is in KB and counts wide coalesced reads at half -> x 1024 x 2; WRITE_SIZE in KB -> x 1024.
r4: further SQ passes may follow (issue / wait split: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY ...; instruction mix:
SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA):
their counters are merged into the same per-kernel records (a counter present in several passes keeps the first pass's value).
usage: make_pmc_json.py trace.db fetch.db write.db sq.db out.json [skip] [more_sq.db ...]"""
import json
import sqlite3
import sys

KERNELS = {   # key -> (substring of the kernel name, skip the first `skip` dispatches?)
    "nn_in_loop": ("nn_stream4", True),          # steady state: dispatches after the first `skip`
    "nn_in_loop_all": ("nn_stream4", False),     # every launch of the fits (what bench.py's live mean launch time covers)
    "nn_bruteforce": ("nn_mfma_kernel", False),
    "blend_fwd": ("panel_gemm3_rb2_kernel", False),
    "blend_bwd": ("panel_gemm3_rb2k", False),
    "blend_wide": ("panel_gemm3_wide", False),
    # r5: the other launches of an iteration (bench.py's roofline.per_kernel)
    "vposer_fwd": ("vposer_fwd", False),
    "vposer_bwd": ("vposer_bwd", False),
    "pose_fwd": ("pose_fwd_kernel", False),
    "pose_bwd": ("pose_bwd_kernel", False),
    "skin_fwd": ("::skin_fwd_kernel", False),
    "blend_skin_fwd": ("blend_skin_fwd_kernel", False),   # r5 (late): the contact set's blend product + skinning in one launch
    "skin_bwd": ("skin_bwd", False),
}


def per_launch(db, sub, skip):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                     "where name like ? group by dispatch_id, counter_name order by dispatch_id", ("%" + sub + "%",)).fetchall()
    ids = sorted({r[0] for r in rows})[skip:]
    keep = set(ids)
    agg, dur = {}, {}
    for d, n, v, t in rows:
        if d in keep:
            agg.setdefault(n, []).append(v)
            dur[d] = t
    out = {n: sum(v) / len(v) for n, v in agg.items()}
    return out, len(ids), (sum(dur.values()) / len(dur) / 1e3 if dur else None)


def trace_avg(db, sub):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, average from top_kernels where name like ?", ("%" + sub + "%",)).fetchall()
    if not rows:
        return None, None, None
    name, calls, avg = max(rows, key=lambda r: r[1])
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return short, calls, avg / 1e3 if avg > 1e4 else avg       # top_kernels reports ns in some builds, us in others


def main(trace_db, fetch_db, write_db, sq_db, out, skip=300, *more):
    skip = int(skip)
    res = {"source": "rocprofv3 passes of tools/run_prof_r5.sh / run_prof_r4.sh on one bench.py command (r5: recorded under `command`; r4: `--steps 1 --warmup 0 "
                     "--no-cpu-baseline --no-logging-run --no-exact-fp32`, two fits), 1 x MI355X; per-dispatch "
                     f"sums over XCD x SE instances, mean over the dispatches after the first {skip} for the in-loop NN kernel "
                     "(steady state), over all dispatches for the others",
           "correction": "gfx950: hbm_bytes = FETCH_SIZE x 1024 x 2 (KB; wide coalesced reads counted at half) + WRITE_SIZE x 1024 (KB)",
           "kernels": {}}
    for key, (sub, steady) in KERNELS.items():
        sk = skip if steady else 0
        sq, n, dur = per_launch(sq_db, sub, sk)
        if not n:
            continue
        f, nf, _ = per_launch(fetch_db, sub, sk)
        w, nw, _ = per_launch(write_db, sub, sk)
        name, calls, avg_us = trace_avg(trace_db, sub)
        k = {"name": name, "dispatches": n, "duration_us_under_pmc": dur, "duration_us_trace": avg_us, "trace_calls": calls}
        k.update(sq)
        for db in more:                                          # further SQ passes: new counters only
            extra, _, _ = per_launch(db, sub, sk)
            for name, val in extra.items():
                k.setdefault(name, val)
        if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
            k["FETCH_SIZE_KB"], k["WRITE_SIZE_KB"] = f["FETCH_SIZE"], w["WRITE_SIZE"]
            k["hbm_bytes"] = 2 * 1024 * f["FETCH_SIZE"] + 1024 * w["WRITE_SIZE"]
        res["kernels"][key] = k
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
