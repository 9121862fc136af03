#!/usr/bin/env python3
"""Work-list cache of the in-loop NN launch over a whole fit: kept / rebuilt waves and list sizes per block of iterations.
Needs the -DFDC_NN_STATS build (FDCAP_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP, first_phase2_iter
from fdcap_amd.io import read_camerapose
C5 = os.environ.get("CONFIG") == "c5"                       # r6: BASELINE config 5 (512 frames, 2 M points, every vertex a contact)
N, ns = int(os.environ.get("FRAMES", "512" if C5 else "1024")), (2000000 if C5 else 500000)
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(ns, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
if C5: l, r = np.arange(0, 5237), np.arange(5237, 10475)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
lib, h = fop.ctx.lib, fop.ctx.handle
raw = ctypes.CDLL(capi.LIB_PATH)
raw.fdcap_debug_nn_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
body = torch.tensor(clip.body_params).cuda()
x78 = torch.empty(N, capi.XDIM, device="cuda")
capi.check(lib.fdcap_params_75_to_78(capi.dptr(body), N, capi.dptr(x78), capi.current_stream()), "75->78")
fop._mode = "global"; fop.init(x78)
P = first_phase2_iter(500)
out = (ctypes.c_ulonglong * 8)()
raw.fdcap_debug_nn_stats(out)
for ii in range(P):
    st = capi.current_stream()
    capi.check(lib.fdcap_opt_backward(h, ii, P, 0, st), "bwd"); capi.check(lib.fdcap_opt_step(h, ii, P, st), "step")
    if ii % 25 == 24 or ii < 3:
        raw.fdcap_debug_nn_stats(out)
        kept, built, rawn, filt, items, mf = out[4], out[5], out[6], out[7], out[3], out[0]
        w = max(kept + built, 1)
        hb = (ctypes.c_ulonglong * 96)(); raw.fdcap_debug_nn_hist(hb)
        if ii % 100 == 99 or ii < 1:
            print("   slow-path entries per wave (log2 buckets):", list(hb[32:48]))
            print("   longest per-lane exact chain per wave (log2 buckets):", list(hb[64:80]))
            print("   work items per wave, log2 buckets [0,1,2-3,4-7,8-15,16-31,32-63,...]: listed", list(hb[:12]), " overflowed", list(hb[16:30]))
        print(f"iter {ii:3d}: kept {kept/w:6.1%}  raw list {rawn/w:5.1f}  after filter {filt/w:5.1f}  work items {items/w:5.1f}  MFMA results/wave {mf/w:5.1f}")
