#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line read from stdin."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
b = r["brute_force"]
print(tag, "frames/s %.1f  ms/step %.1f  in-loop NN %.3f ms (alg %.0f GB/s = %.2f x HBM peak)  brute force %.2f ms (%.1f%% HBM-alg, %.2e pairs/s)" % (
    d["value"], d["ms_per_step"], r["ms_per_launch"], r["achieved"], r["frac"], b["ms_per_launch"], 100 * b["frac"],
    b["compute_side"]["pairs_per_s"]))
if "steady_state_ms_per_launch" in r:
    print("   in-loop NN: mean of %d launches %.4f ms, steady state %.4f ms" % (r.get("launches_timed", 0), r["ms_per_launch"], r["steady_state_ms_per_launch"]))
if d.get("with_reference_logging"):
    print("   with every loss term every iteration: %.1f frames/s" % d["with_reference_logging"]["value"])
if "blendshape_gemm" in d:
    g = d["blendshape_gemm"]
    print("   blend GEMM %.3f ms  %.1f TFLOP/s (%.1f%% of fp32 MFMA peak)" % (g["ms_per_launch"], g["achieved"], 100 * g["frac"]))
    if "executed" in g:
        print("      executed: %.0f bf16 TFLOP/s = %.1f%% of the dense bf16 MFMA peak" % (g["executed"]["achieved"], 100 * g["executed"]["frac"]))
