#!/usr/bin/env python3
"""ops.BodyModel + ops.VPoser forward / backward in a torch loop (the operator-by-operator integration path) at clip size, for
rocprofv3 --kernel-trace + tools/trace_outliers.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1)
ctx = capi.Context(bm, vp)
body = ops.BodyModel(ctx); vposer = ops.VPoser(ctx)
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn(B, 32, device="cuda", generator=g, requires_grad=True)
go = (0.3 * torch.randn(B, 3, device="cuda", generator=g)).requires_grad_(True)
betas = (0.5 * torch.randn(B, 10, device="cuda", generator=g)).requires_grad_(True)
lh = (0.2 * torch.randn(B, 12, device="cuda", generator=g)).requires_grad_(True)
rh = (0.2 * torch.randn(B, 12, device="cuda", generator=g)).requires_grad_(True)
tr = torch.randn(B, 3, device="cuda", generator=g).requires_grad_(True)
for k in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bp = vposer.decode(z, "aa").reshape(B, -1)
    out = body(return_verts=True, body_pose=bp, transl=tr, global_orient=go, betas=betas, left_hand_pose=lh, right_hand_pose=rh)
    loss = out.vertices.square().mean() + out.joints.square().mean()
    loss.backward()
    torch.cuda.synchronize()
    if k in (0, 1, 11): print("iteration %d: %.3f ms" % (k, 1e3 * (time.perf_counter() - t0)))
ctx.close()
