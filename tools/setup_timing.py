#!/usr/bin/env python3
"""What the untimed part of a run costs (VERDICT r5 "what's weak" 2): seconds in fdcap_ctx_create, fdcap_set_scene,
fdcap_set_contact_ids, the first fit (fdcap_opt_create + the seeding launch) next to a steady fit, per BASELINE configuration.
usage: python tools/setup_timing.py [c3 c2 c5 cli300]   (cli300: 300 frames, 500 k points -- the reference's real clip length)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

SIZES = {"c3": (1024, 500_000, False), "c5": (512, 2_000_000, True), "c2": (256, 100_000, False), "cli300": (300, 500_000, False)}


def measure(name, iters=500):
    import fdcap_amd  # noqa: F401
    from fdcap_amd import capi, synth
    from fdcap_amd.fitting import FittingOP
    from fdcap_amd.io import read_camerapose
    f, ns, allc = SIZES[name]
    bm = synth.make_body_model(10475, seed=0, lbs_nnz=4)
    vp = synth.make_vposer(seed=1)
    clip = synth.make_clip(f, seed=3)
    scene = synth.make_scene(ns, seed=2)
    left, right = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
    vid = np.arange(10475) if allc else np.concatenate([left, right])
    cam = read_camerapose(clip.camerapose_lines)
    torch.cuda.synchronize()
    out = {"frames": f, "scene_points": ns, "contact_verts": int(len(vid))}
    t0 = time.perf_counter()
    ctx = capi.Context(bm, vp)
    torch.cuda.synchronize()
    out["ctx_create_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.set_scene(scene)
    torch.cuda.synchronize()
    out["set_scene_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.set_contact_ids(vid)
    torch.cuda.synchronize()
    out["set_contact_ids_s"] = time.perf_counter() - t0
    ctx.close()
    # the same through the host class, then fits
    t0 = time.perf_counter()
    fop = FittingOP({"num_iter": iters}, {}, f, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid, camera_ext=cam)
    torch.cuda.synchronize()
    out["fittingop_init_s"] = time.perf_counter() - t0
    body = torch.tensor(clip.body_params).cuda()
    fits = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b, sc, cm = fop.fitting(body, "global")
        b = b.cpu(); cm = cm.cpu()
        torch.cuda.synchronize()
        fits.append(time.perf_counter() - t0)
    out["first_fit_s"], out["steady_fit_s"] = fits[0], min(fits[1:])
    out["cli_end_to_end_frames_per_s"] = f / (out["fittingop_init_s"] + fits[0])
    out["steady_frames_per_s"] = f / out["steady_fit_s"]
    fop.close()
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or ["c3", "c2", "c5", "cli300"]
    torch.zeros(1).cuda()
    res = {}
    for n in names:
        res[n] = measure(n)
        print(n, json.dumps(res[n]), flush=True)
