#!/usr/bin/env python3
"""Development tool: does any kernel read LDS or registers it never wrote?  Runs fits with the product library and with the
-DFDC_POISON build (every launch preceded by one that leaves NaN patterns in LDS and in the vector registers; build it with
tools/build_variant.sh poison -DFDC_POISON) and compares the results bit for bit -- single rank and two ranks sharing the GPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.multiprocessing as mp


def _single(q, mode, n, iters):
    from tests.test_gpu_sharded import _fit
    q.put(_fit(None, mode, n, iters))


def single(mode, n, iters, lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    old = os.environ.get("FDCAP_LIB")
    if lib: os.environ["FDCAP_LIB"] = lib
    try:
        p = ctx.Process(target=_single, args=(q, mode, n, iters)); p.start(); r = q.get(timeout=600); p.join(timeout=60)
    finally:
        if lib:
            if old is None: os.environ.pop("FDCAP_LIB", None)
            else: os.environ["FDCAP_LIB"] = old
    return r


def eq(a, b):
    return (np.array_equal(a[1], b[1]), a[2] == b[2], np.array_equal(a[3], b[3]), np.array_equal(a[4], b[4]))


if __name__ == "__main__":
    from tests.test_gpu_sharded import _run_ranks
    P = os.path.join(ROOT, "4dcapture-fpv_amd", "libfdcap_hip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "poison"))
    assert os.path.exists(P), "build the poison variant first"
    for mode, n, iters in (("global", 200, 10), ("local", 40, 10), ("global", 22, 10)):
        a = single(mode, n, iters, None); b = single(mode, n, iters, P)
        print(f"single rank {mode} n={n}: product vs poisoned (body, scale, cam, totals) equal: {eq(a, b)}", flush=True)
    for ov in ("0", "1"):
        a = _run_ranks(2, "global", 200, ov, 10)
        os.environ["FDCAP_LIB"] = P
        b = _run_ranks(2, "global", 200, ov, 10)
        os.environ.pop("FDCAP_LIB")
        print(f"two ranks, overlap {ov}: product vs poisoned equal per rank:", [eq(x[1:], y[1:]) for x, y in zip(a, b)], flush=True)
