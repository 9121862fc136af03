#!/usr/bin/env python3
"""Instrumented run of the MFMA-filtered NN kernel (build with -DFDC_NN_STATS): how often the
exact path is taken on the bench workload.  Development tool, not part of the product."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libfdcap_hip_stats.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DFDC_BUILD_NO_PK_F32", "-DFDC_NN_STATS", "-o", lib,
                       os.path.join(ROOT, "4dcapture-fpv_amd", "csrc", "fdcap.hip")])
os.environ["FDCAP_LIB"] = lib
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N, ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 500000
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(ns, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": int(os.environ.get("NN_ITERS", "2"))}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
fop.fitting(torch.tensor(clip.body_params).cuda(), "global")
L = capi.load_library()
out = (ctypes.c_ulonglong * 4)()
L.fdcap_debug_nn_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
L.fdcap_debug_nn_stats(out)
ms = ctypes.c_float()
capi.check(L.fdcap_opt_time_chamfer(fop.ctx.handle, 1, int(os.environ.get("NN_BRUTE", "0")), ctypes.byref(ms), capi.current_stream()), "time")
L.fdcap_debug_nn_stats(out)   # one warm-up + one timed launch
tot, slow, rows = max(out[0], 1), out[1], out[2]
print('chunks staged / visited by waves', out[3], 'of', 2 * ((N * 500 + 511) // 512) * ((ns + 511) // 512), '(staged kernel) ;', out[3] / 2 / (N * 500 / 64), 'per 64-query group per launch')
print(f"ms/launch {ms.value:.3f}  MFMA results {tot}  exact-path entries {slow} ({100.0*slow/tot:.2f} %)  rows re-evaluated {rows} "
      f"({rows/max(slow,1):.2f} per entry, {rows/(2*N*500):.1f} per query-launch)  [2 launches]")
