#!/usr/bin/env python3
"""The three operators of the operator-by-operator boundary (ops.BodyModel, ops.VPoser, ops.chamferDist against the registered
scene) over batch sizes: milliseconds per forward + backward.  r6: looks for form-selection cliffs (non-monotone steps in B) in the
paths the optimiser loop does not take.   usage: python tools/op_sweep.py [B ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, ops, synth
Bs = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 192, 256, 300, 384, 512, 768, 1024]
bm = synth.make_body_model(10475, seed=0)
ctx = capi.Context(bm, synth.make_vposer(seed=1))
scene = synth.make_scene(100_000, seed=2)
ctx.set_scene(scene)
s_dev = torch.tensor(scene, device="cuda")
body, vposer, cham = ops.BodyModel(ctx), ops.VPoser(ctx), ops.chamferDist(ctx, both=False)
def timed(fn, reps=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for B in Bs:
    g = torch.Generator(device="cuda").manual_seed(B)
    z = (0.5 * torch.randn(B, 32, device="cuda", generator=g)).requires_grad_(True)
    go = (0.3 * torch.randn(B, 3, device="cuda", generator=g)).requires_grad_(True)
    be = (0.5 * torch.randn(B, 10, device="cuda", generator=g)).requires_grad_(True)
    lh = (0.2 * torch.randn(B, 12, device="cuda", generator=g)).requires_grad_(True)
    tr = torch.randn(B, 3, device="cuda", generator=g).requires_grad_(True)
    def f_body():
        aa = vposer.decode(z, output_type="aa").view(B, -1)
        out = body(return_verts=True, body_pose=aa, transl=tr, global_orient=go, betas=be, left_hand_pose=lh, right_hand_pose=lh)
        (out.vertices.sum() + out.joints.sum()).backward()
    def f_vp():
        vposer.decode(z, output_type="aa").sum().backward()
    q = (s_dev[torch.randint(0, len(scene), (B, 500), device="cuda", generator=g)] + 0.02 * torch.randn(B, 500, 3, device="cuda", generator=g)).requires_grad_(True)
    def f_ch():
        d, _ = cham(q, s_dev.unsqueeze(0).expand(B, -1, -1))
        d.sum().backward()
    print(f"B {B:5d}: VPoser + body model (full mesh) fwd+bwd {timed(f_body):7.3f} ms | VPoser alone {timed(f_vp):6.3f} ms | Chamfer 500 queries/frame vs 100 k {timed(f_ch):6.3f} ms", flush=True)
