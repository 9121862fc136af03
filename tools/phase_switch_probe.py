#!/usr/bin/env python3
"""What the in-loop Chamfer search does launch by launch around the phase switch of an every-iteration-logging fit (-DFDC_NN_STATS
build): waves on a kept list / waves that built one, list items before and after the per-query filter, MFMA results, exact-path
entries -- and the launch's duration (HIP events around the iteration).  Development tool.
usage: phase_switch_probe.py [first iteration] [last iteration]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libfdcap_hip_stats.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DFDC_BUILD_NO_PK_F32",
                       "-DFDC_NN_STATS", "-o", lib, os.path.join(ROOT, "4dcapture-fpv_amd", "csrc", "fdcap.hip")])
os.environ["FDCAP_LIB"] = lib
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
lo, hi = (int(sys.argv[1]) if len(sys.argv) > 1 else 396), (int(sys.argv[2]) if len(sys.argv) > 2 else 440)
N, ns = int(os.environ.get("FRAMES", "1024")), int(os.environ.get("SCENE", "500000"))       # (FRAMES=512 SCENE=2000000 ALLC=1: BASELINE config 5)
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(ns, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.arange(10475) if os.environ.get("ALLC") == "1" else np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
L = capi.load_library()
L.fdcap_debug_nn_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 8)()
L.fdcap_debug_nn_hist.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
hist = (ctypes.c_ulonglong * 96)()
import time
t_last = [time.perf_counter()]
print("iter   wall_us  kept_waves built_waves raw_items kept_items  mfma_results exact_entries rows")
def hook(k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    L.fdcap_debug_nn_stats(out)
    print(f"{k:4d} {1e6 * (t - t_last[0]):9.0f} {out[4]:10d} {out[5]:10d} {out[6]:9d} {out[7]:10d} {out[0]:12d} {out[1]:12d} {out[2]:8d}")
    if os.environ.get("HIST") == "1":                     # waves by log2(work items) since the last look (bins: 0, 1, 2-3, 4-7, ...)
        L.fdcap_debug_nn_hist(hist)
        print("      waves by work items (0 | 1 | 2-3 | 4-7 | 8-15 | 16-31 | 32-63 | 64-127 | 128-255 | ...):", " ".join(str(hist[i]) for i in range(12)))
    t_last[0] = time.perf_counter()
fop.snapshot_hook = hook
fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1, snapshot_at=list(range(lo, hi)) + [int(v) for v in os.environ.get("EXTRA", "1,2,3,10,11,50,51,200,201").split(",")])
