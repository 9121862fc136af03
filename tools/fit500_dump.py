#!/usr/bin/env python3
"""Runs the HIP fit on tests/golden/ref_global_500it.npz's inputs (the reference's own 500-iteration run) and writes what came out
to gpurun_out/fit500_hip.npz: snapshots at the fixture's iterations, the per-iteration log, the final triple.  The distances are
then computed wherever the oracle can run (tests/parity500.py); tests/test_gpu_parity500.py is the committed form of the check."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fdcap_amd  # noqa: E402,F401
from fdcap_amd import synth  # noqa: E402
from fdcap_amd.fitting import FittingOP  # noqa: E402
from fdcap_amd.io import read_camerapose  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "ref_global_500it.npz"))
bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
vp = synth.make_vposer(seed=int(g["vposer_seed"]))
out = {}
TAG = sys.argv[1] if len(sys.argv) > 1 else "split3"       # (FDCAP_GEMM_SPLIT3=0 python tools/fit500_dump.py fp32: exact-fp32 products)
for tag in ("hip", ):
    fop = FittingOP({"num_iter": 500}, {}, 300, body_model=bm, vposer=vp, scene_verts=g["scene"], contact_ids=g["vid"],
                    camera_ext=read_camerapose(list(g["camerapose"])))
    body, scale, cam = fop.fitting(torch.tensor(g["body_in"]).cuda(), "global", log_every=1, snapshot_at=[int(k) for k in g["snap_iters"]])
    ks = sorted(fop.snapshots)
    out[f"{tag}_snap_iters"] = np.array(ks)
    out[f"{tag}_snap_x78"] = np.stack([fop.snapshots[k][0].cpu().numpy() for k in ks])
    out[f"{tag}_snap_scale"] = np.array([float(fop.snapshots[k][1].cpu()) for k in ks])
    out[f"{tag}_snap_cam"] = np.stack([fop.snapshots[k][2].cpu().numpy() for k in ks])
    lg = fop.log
    out[f"{tag}_log"] = np.array([lg.iters, lg.l_rec, lg.l_vposer, lg.loss_smoothing, lg.loss_contact, lg.loss_world_smoothing, lg.total]).T
    out[f"{tag}_body_rec"] = body.cpu().numpy()
    out[f"{tag}_scale"] = np.float32(scale)
    out[f"{tag}_cam"] = cam.cpu().numpy()
    out[f"{tag}_idx1"] = np.asarray(fop.idx1)
    fop.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fit500_{TAG}.npz"), **out)
print(f"wrote gpurun_out/fit500_{TAG}.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
