#!/usr/bin/env python3
"""Upper bound for running the clip as TWO half-clips on two streams of one GPU (r4 question: do two independent latency-bound
chains fill each other's idle phases?).  Two UNCOUPLED 512-frame fits (no halo / scale exchange: an upper bound for the coupled
form) issued from two host threads on two streams, against the same two fits one after the other and one 1024-frame fit."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose


def make(n, seed):
    bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(n, seed=seed)
    scene = synth.make_scene(500_000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
    fop = FittingOP({"num_iter": 500}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                    camera_ext=read_camerapose(clip.camerapose_lines))
    return fop, torch.tensor(clip.body_params).cuda()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


big, xb = make(1024, 3)
print(f"one 1024-frame fit: {timed(lambda: big.fitting(xb, 'global')):.1f} ms")
big.close()
halves = [make(512, 3), make(512, 5)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(k):
    with torch.cuda.stream(streams[k]):
        halves[k][0].fitting(halves[k][1], "global")


def seq():
    run(0); run(1)


def par():
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()


print(f"two 512-frame fits, one after the other: {timed(seq):.1f} ms")
print(f"two 512-frame fits, two threads / two streams: {timed(par):.1f} ms")
