#!/usr/bin/env python3
"""One fit in mode 'local' and one in mode 'dct' at clip size (run under rocprofv3 --kernel-trace; tools/trace_outliers.py reads the
result): are there launches far above their kernel's median in the modes the bench does not time?
   usage: modes_trace.py [local|dct] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
mode = sys.argv[1] if len(sys.argv) > 1 else "local"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
cfg = {"num_iter": 100} if mode == "local" else {"num_iter": 100, "dct_num_iter": 2000}
fop = FittingOP(cfg, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
body = torch.tensor(clip.body_params).cuda()
for k in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fop.fitting(body, mode)
    torch.cuda.synchronize(); print("mode %s, %d frames, fit %d: %.2f ms" % (mode, N, k, 1e3 * (time.perf_counter() - t0)))
