#!/usr/bin/env python3
"""Phases of the Chamfer search's waves (set-up | list | filter | main loop | tail; -DFDC_NN_TIMELINE build, built here) for the
launches around the phase switch of an every-iteration-logging fit: WHICH part of a wave's life grows when camera_ext starts to move.
usage: phase_switch_timeline.py [iterations ...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libfdcap_hip_tl.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DFDC_BUILD_NO_PK_F32",
                       "-DFDC_NN_TIMELINE"] + [a for a in sys.argv[1:] if a.startswith("-D")] + ["-o", lib, os.path.join(ROOT, "4dcapture-fpv_amd", "csrc", "fdcap.hip")],
                      stderr=subprocess.DEVNULL)
os.environ["FDCAP_LIB"] = lib
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
its = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [2, 10, 100, 399, 400, 401, 402, 403, 404, 405, 408, 415, 430, 460, 499]
N, ns = 1024, 500000
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(ns, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
raw = ctypes.CDLL(capi.LIB_PATH)
CAP = 16384
nb = min(CAP, ((N * 500 + 31) // 32 + 7) // 8 * 8)
buf = (ctypes.c_ulonglong * (CAP * 8))()
names = ["set-up", "list", "filter", "main", "tail"]
print("iter  span_us  lifetime q50/q90/max | " + " | ".join(f"{n} q50/q90" for n in names) + " | items q50/q90/max")
def hook(k):
    torch.cuda.synchronize()
    assert raw.fdcap_debug_nn_timeline(buf, CAP * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(CAP, 8)[:nb].astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    ph = np.stack([a[:, 3] - a[:, 0], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4], a[:, 6] - a[:, 5], a[:, 1] - a[:, 6]], axis=1) / 100.0
    q = lambda x, p: float(np.quantile(x, p))
    print(f"{k:4d} {en.max():8.1f}  {q(en-st,.5):5.1f}/{q(en-st,.9):5.1f}/{(en-st).max():6.1f} | " +
          " | ".join(f"{q(ph[:, i],.5):5.2f}/{q(ph[:, i],.9):6.2f}" for i in range(5)) + f" | {q(a[:,7],.5):.0f}/{q(a[:,7],.9):.0f}/{a[:,7].max()}", flush=True)
fop.snapshot_hook = hook
fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1, snapshot_at=its)
