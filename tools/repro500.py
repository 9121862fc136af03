#!/usr/bin/env python3
"""Race screen: three full 500-iteration fits of BASELINE config 3 must give identical bits (python tools/repro500.py)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import fdcap_amd
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N=1024
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
outs=[]
for k in range(3):
    fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]), camera_ext=read_camerapose(clip.camerapose_lines))
    body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "global")
    outs.append((body.clone(), float(scale), cam.clone())); fop.close()
for k in (1,2):
    print("run", k, "equal to run 0:", torch.equal(outs[0][0], outs[k][0]), outs[0][1] == outs[k][1], torch.equal(outs[0][2], outs[k][2]))
print("finite:", bool(torch.isfinite(outs[0][0]).all()), "scale", outs[0][1])
