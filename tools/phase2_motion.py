#!/usr/bin/env python3
"""How far do the bodies move per iteration in phase 1 and in phase 2 (camera_ext is optimised there)?  Bench workload,
snapshots of consecutive iterations; prints the per-iteration change of the parameters that move the whole body."""
import os, sys
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
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
ks = [200, 201, 300, 301, 390, 391, 402, 403, 404, 420, 421, 450, 451, 498, 499]
fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1, snapshot_at=ks)
for a in (200, 300, 390, 402, 403, 420, 450, 498):
    xa, sa, ca = fop.snapshots[a]; xb, sb, cb = fop.snapshots[a + 1]
    dt = (xb[:, :3] - xa[:, :3]).norm(dim=1)                       # transl (body frame, before scale)
    dc = (cb.view(-1, 4, 4)[:, :3, 3] - ca.view(-1, 4, 4)[:, :3, 3]).norm(dim=1)
    dR = (cb.view(-1, 4, 4)[:, :3, :3] - ca.view(-1, 4, 4)[:, :3, :3]).flatten(1).norm(dim=1)
    print(f"iter {a}->{a+1}: |d transl| median {dt.median()*1e3:.2f} mm  |d cam t| median {dc.median()*1e3:.2f} mm  |d cam R|_F median {dR.median():.4f}  d scale {float(sb-sa):+.5f}")
