#!/usr/bin/env python3
"""optimization.py's per-frame smoother over a whole clip (fdcap_frame_smoother, one launch): time per clip and per Adam step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import synth, smoother
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
clip = synth.make_clip(N, seed=3)
for k in range(3):
    op = smoother.FittingOP()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = op.fitting_clip(clip.body_params)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("smoother, %d frames x %d steps: %.2f ms = %.3f us per Adam step" % (N, op.num_iter, 1e3 * dt, 1e6 * dt / (N * op.num_iter)))
