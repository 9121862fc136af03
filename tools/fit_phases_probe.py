import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N = 1024
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
body = torch.tensor(clip.body_params).cuda()
orig_init = fop.init
def timed_init(x):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_init(x)
    torch.cuda.synchronize(); print("   init %.3f ms" % (1e3 * (time.perf_counter() - t0)))
    return r
fop.init = timed_init
for k in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fop.fitting(body, "global")
    torch.cuda.synchronize(); print("fit %d: %.3f ms" % (k, 1e3 * (time.perf_counter() - t0)))
