#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line read from stdin."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, "frames/s %.1f  ms/step %.2f" % (d["value"], d["ms_per_step"]))
if d.get("with_reference_logging"):
    w = d["with_reference_logging"]
    print("   every loss term every iteration: %.1f frames/s  (%.2f ms/step)" % (w["value"], w["ms_per_step"]))
if d.get("exact_fp32") and d["exact_fp32"].get("value"):
    print("   exact fp32 chains: %.1f frames/s (%.2f ms/step)" % (d["exact_fp32"]["value"], d["exact_fp32"]["ms_per_step"]))
r = d.get("roofline")
if r:
    print("   in-loop NN: mean of %d launches %.4f ms, steady state %.4f ms" % (r.get("launches_timed", 0), r["ms_per_launch"], r["steady_state_ms_per_launch"]))
    if r.get("frac") is not None:
        print("      bound %s: frac %.3f (%s)  hbm frac %s  mfma busy %s  waves/SIMD %s" % (
            r["bound"], r["frac"], r["unit"], r.get("hbm_frac_on_counter_bytes"), r.get("mfma_busy_frac"), r.get("mean_waves_per_simd")))
    c = r["contract"]
    print("      contract: %.0f GB/s on algorithmic bytes = %.2f x HBM peak" % (c["achieved"], c["frac_on_algorithmic_bytes"]))
    b = r["brute_force"]
    print("      brute force %.2f ms: %.0f TFLOP/s bf16 = %.3f of dense peak; %.1f%% HBM on algorithmic bytes" % (
        b["ms_per_launch"], b["achieved"], b["frac"], 100 * b["hbm"]["frac"]))
g = d.get("blendshape_gemm")
if g:
    print("   blend GEMM %.3f ms  %.0f %s = frac %.3f" % (g["ms_per_launch"], g["achieved"], g["unit"], g["frac"]))
    for k, v in (g.get("in_loop") or {}).items():
        if isinstance(v, dict):
            print("      %s %.2f us frac %.3f" % (k, v["us_per_launch"], v["frac"]))
c = d.get("cpu_baseline")
if c:
    print("   cpu baseline %.5f frames/s on %d threads (%d frames sampled)" % (c["value"], c["cores"], c.get("sample_frames", 0)))
