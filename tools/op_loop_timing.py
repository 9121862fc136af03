#!/usr/bin/env python3
"""The reference-shaped call pattern of Op 1 at BASELINE config 2 sizes (256 frames x 500 contact vertices vs a 100 k-point scene,
the same scene tensor every call, queries drifting a few millimetres per call): milliseconds per ops.chamferDist forward + backward
through the generic every-pair scan and through the registered-scene search.  Development tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, ops, synth
B, n, ns = int(os.environ.get("FRAMES", 256)), 500, int(os.environ.get("SCENE", 100_000))
bm = synth.make_body_model(300, seed=0)
ctx = capi.Context(bm, synth.make_vposer(seed=1))
scene = synth.make_scene(ns, seed=2)
ctx.set_scene(scene)
s_batch = torch.tensor(scene, device="cuda").unsqueeze(0).expand(B, -1, -1)
rng = np.random.default_rng(0)
x0 = torch.tensor(scene[rng.integers(0, ns, size=(B, n))] + rng.normal(0, 0.02, size=(B, n, 3)).astype(np.float32), device="cuda")
for name, op in (("generic every-pair scan (nn_mfma_kernel)", ops.chamferDist(ctx, both=False, use_registered_scene=False)),
                 ("registered scene (culled, seeded: nn_stream4_kernel)", ops.chamferDist(ctx, both=False))):
    x = x0.clone()
    for it in range(60):
        if it == 10:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        x = (x + 0.002 * torch.randn_like(x)).detach().requires_grad_(True)
        d, _ = op(x, s_batch)
        d.sum().backward()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per forward + backward  ({B} x {n} queries vs {ns} points)")
