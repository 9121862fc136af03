#!/usr/bin/env python3
"""Are two builds of the library bit-identical in what they compute?  Runs the same fits (a small one with every loss term logged,
mode 'local', 60 iterations of the bench workload, and a short fit with all 10 475 vertices as contacts) once per library in a child process each and compares parameters, scale,
camera_ext and the loss log bit for bit.   tools/compare_builds.py <libA.so> <libB.so>
r6: an argument of the form NAME=value (instead of a path) runs the tree's library with that environment setting, so two
settings of an A/B switch can be compared the same way:   tools/compare_builds.py FDCAP_PN_WIDE_RB=2 FDCAP_PN_WIDE_RB=4"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, dataclasses, numpy as np, torch
sys.path.insert(0, %r)
import fdcap_amd
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
out = {}
def fit(tag, n, V, ns, per_part, iters, mode, log_every):
    bm = synth.make_body_model(V, seed=7); vp = synth.make_vposer(seed=8); clip = synth.make_clip(n, seed=9)
    scene = synth.make_scene(ns, seed=10); l, r = synth.make_contact_ids(bm.v_template, per_part=max(per_part, 1), seed=11)
    if per_part == 0: l, r = np.arange(0, V // 2), np.arange(V // 2, V)        # every vertex a contact (BASELINE config 5's forms)
    fop = FittingOP({"num_iter": iters}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                    camera_ext=read_camerapose(clip.camerapose_lines), n_left=len(l))
    b, s, c = fop.fitting(torch.tensor(clip.body_params).cuda(), mode, log_every=log_every)
    out[tag + "_body"] = b.cpu().numpy(); out[tag + "_scale"] = np.float64(s); out[tag + "_cam"] = c.cpu().numpy()
    if log_every:
        for k, v in dataclasses.asdict(fop.log).items(): out[tag + "_log_" + k] = np.asarray(v, dtype=np.float64)
    fop.close()
fit("small", 37, 300, 6000, 20, 40, "global", 1)
fit("local", 21, 300, 6000, 20, 30, "local", 1)
fit("bench", 1024, 10475, 500000, 250, 60, "global", 0)
fit("allverts", 24, 10475, 50000, 0, 12, "global", 1)
np.savez(sys.argv[1], **out)
'''
res = []
for lib in sys.argv[1:3]:
    f = tempfile.mktemp(suffix=".npz")
    if "=" in lib and not os.path.exists(lib):
        k, v = lib.split("=", 1)
        env = dict(os.environ, **{k: v})
    else:
        env = dict(os.environ, FDCAP_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", CHILD % ROOT, f], env=env, check=True)
    import numpy as np
    res.append(dict(np.load(f)))
bad = 0
for k in sorted(res[0]):
    a, b = res[0][k], res[1][k]
    same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    if not same:
        bad += 1
        print(f"DIFFERENT {k}: max |d| = {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64))):.3e}")
print(f"{len(res[0])} arrays compared, {bad} differ" + ("" if bad else ": the two builds are bit-identical on these fits"))
sys.exit(1 if bad else 0)
