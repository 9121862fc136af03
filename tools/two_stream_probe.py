#!/usr/bin/env python3
"""Probe (development tool): how much faster do two independent 512-frame fits run on two HIP streams of one process than
one 1024-frame fit?  Upper bound for splitting a clip into in-process shards (no exchange here)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose

def make(n, seed):
    bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(n, seed=seed)
    scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
    fop = FittingOP({"num_iter": 500}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                    camera_ext=read_camerapose(clip.camerapose_lines))
    return fop, torch.tensor(clip.body_params).cuda()

def run(fop, x, stream):
    with torch.cuda.stream(stream):
        fop.fitting(x, "global")
    stream.synchronize()

full, xf = make(1024, 3)
s0 = torch.cuda.Stream()
run(full, xf, s0)
t = time.perf_counter(); run(full, xf, s0); t_full = time.perf_counter() - t
full.close()
a, xa = make(512, 3); b, xb = make(512, 5)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
run(a, xa, s1); run(b, xb, s2)
t = time.perf_counter(); run(a, xa, s1); run(b, xb, s2); t_seq = time.perf_counter() - t
t = time.perf_counter()
th = [threading.Thread(target=run, args=(a, xa, s1)), threading.Thread(target=run, args=(b, xb, s2))]
[x.start() for x in th]; [x.join() for x in th]
t_par = time.perf_counter() - t
print(f"one 1024-frame fit {t_full*1e3:.1f} ms; two 512-frame fits back to back {t_seq*1e3:.1f} ms; on two streams / two threads {t_par*1e3:.1f} ms")
