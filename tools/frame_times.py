#!/usr/bin/env python3
"""Per-frame phase times (s_memtime) of pose_fwd_kernel / pose_bwd_kernel in the loop; needs the -DFDC_PN_TIMING build (FDCAP_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N = int(os.environ.get("FRAMES", "1024"))          # (FRAMES=128 SCENE=2000000 ALLC=1: BASELINE config 5's kernels, 11 chunk workgroups per frame)
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(int(os.environ.get("SCENE", "500000")), seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 40}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.arange(10475) if os.environ.get("ALLC") == "1" else np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines))
body = torch.tensor(clip.body_params).cuda()
lib, h = fop.ctx.lib, fop.ctx.handle
x78 = torch.empty(N, capi.XDIM, device="cuda")
capi.check(lib.fdcap_params_75_to_78(capi.dptr(body), N, capi.dptr(x78), capi.current_stream()), "75->78")
fop._mode = "global"; fop.init(x78)
for ii in range(20):
    capi.check(lib.fdcap_opt_backward(h, ii, 10**6, 0, capi.current_stream()), "bwd"); capi.check(lib.fdcap_opt_step(h, ii, 10**6, capi.current_stream()), "step")
raw = ctypes.CDLL(capi.LIB_PATH)
buf = (ctypes.c_ulonglong * (3 * 2048 * 8))()
assert raw.fdcap_debug_frame_times(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(3, 2048, 8).astype(np.int64)
names = [("pose_fwd", ["sum partials + topology", "joint rotations + J", "chain", "outputs"], 5),
         ("pose_bwd", ["stage + param-loss grads", "load + own grads", "products + reverse chain", "dR/drel", "rot backward", "tail reductions"], 7),
         ("skin_bwd", ["stage A", "vertex loop", "dA reduction", "final sums + stores"], 5)]
for k, (nm, ph, ns) in enumerate(names):
    t = a[k, :, :ns]
    t = t[t[:, -1] > 0]
    print(nm, len(t), "frames; lifetime q50", int(np.median(t[:, -1] - t[:, 0])), "cycles")
    for i in range(ns - 1):
        d = t[:, i + 1] - t[:, i]
        print(f"   {ph[i]:32s} q50 {int(np.median(d)):6d}  q90 {int(np.quantile(d, 0.9)):6d}")
    if nm == "pose_bwd":
        tt = a[k, :N, :]; tt = tt[tt[:, 6] > 0]
        if (tt[:, 7] > 0).all():
            print(f"   (of the first phase: staging batch until the barrier q50 {int(np.median(tt[:, 7] - tt[:, 0]))} cycles)")

if not hasattr(raw, "fdcap_debug_gemm_times"):
    sys.exit(0)
gb = (ctypes.c_ulonglong * (2 * 8192 * 4))()
assert raw.fdcap_debug_gemm_times(gb) == 0
g = np.frombuffer(gb, dtype=np.uint64).reshape(2, 8192, 4).astype(np.int64)
for k, nm in enumerate(["panel_gemm fwd (K=496, N=1500)", "panel_gemm bwd (K=1500, N=496)"]):
    t = g[k]; t = t[t[:, 3] > 0]
    print(nm, len(t), "WGs; lifetime q50", int(np.median(t[:, 3] - t[:, 0])), "max", int((t[:, 3] - t[:, 0]).max()), "cycles")
    for i, ph in enumerate(["stage A (+ first fragments)", "mma", "store"]):
        d = t[:, i + 1] - t[:, i]
        print(f"   {ph:32s} q10 {int(np.quantile(d, 0.1)):6d} q50 {int(np.median(d)):6d}  q90 {int(np.quantile(d, 0.9)):6d}")
