#!/usr/bin/env python3
"""Development tool: is a SINGLE-rank fit still bit-reproducible while another process keeps the same GPU busy?
(Two-rank tests on a one-GPU box show rare last-bit differences in one frame; this separates "two processes share the GPU"
from "the sharded schedule".)  Usage: contention_probe.py [frames] [reps] [load: none|matmul|fit]"""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LOAD = r'''
import sys, time, torch
sys.path.insert(0, %r)
kind = sys.argv[1]
if kind == "matmul":
    a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20): c = a @ b
        torch.cuda.synchronize()
else:
    from tests.test_gpu_sharded import _fit
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        _fit(None, "global", 200, 10)
''' % ROOT

FIT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_sharded import _fit
r = _fit(None, "global", int(sys.argv[1]), 10)
np.savez(sys.argv[2], body=r[1], scale=r[2], cam=r[3], tot=r[4])
''' % ROOT

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    load = sys.argv[3] if len(sys.argv) > 3 else "matmul"
    bg = None
    if load != "none":
        bg = subprocess.Popen([sys.executable, "-c", LOAD, load, str(20 + 6 * reps)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        time.sleep(8)
    outs = []
    for k in range(reps + 1):
        f = f"/tmp/contention_{k}.npz"
        subprocess.run([sys.executable, "-c", FIT, str(n), f], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        outs.append(np.load(f))
    bad = 0
    for k in range(1, reps + 1):
        same = all(np.array_equal(outs[0][key], outs[k][key]) for key in ("body", "scale", "cam", "tot"))
        if not same:
            bad += 1
            rows = sorted(set(np.argwhere(outs[0]["body"] != outs[k]["body"])[:, 0].tolist()))
            print(f"  run {k}: body rows differing {rows[:10]}, totals equal {np.array_equal(outs[0]['tot'], outs[k]['tot'])}", flush=True)
    print(f"single rank n={n}, load {load}: {bad} of {reps} runs differ from the first", flush=True)
    if bg is not None:
        bg.terminate(); bg.wait()
