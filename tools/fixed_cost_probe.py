#!/usr/bin/env python3
"""What of a step is not the loop?  Bench workload with num_iter = 500 / 100 / 2: the intercept of the line through the
three times is the fixed cost of a fit (init: outlier search on the host, 75 -> 78, seeds; results: 78 -> 75, D -> H)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N = 1024
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500_000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
x = torch.tensor(clip.body_params).cuda()
for it in (500, 100, 2):
    fop = FittingOP({"num_iter": it}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                    camera_ext=read_camerapose(clip.camerapose_lines))
    def step():
        b, s, c = fop.fitting(x, "global"); return b.cpu(), s, c.cpu()
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"num_iter {it}: {min(ts):.2f} ms (min of 5), median {sorted(ts)[2]:.2f}")
    if it == 2:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    fop.close()
