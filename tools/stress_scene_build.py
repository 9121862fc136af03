#!/usr/bin/env python3
"""Randomised stress of fdcap_set_scene's device build against the host specification (FDCAP_SCENE_BUILD=host): 222 scenes of 1 .. 1 M
points -- uniform, anisotropic, coordinates on a coarse grid (ties across every cut), duplicated points, the synthetic room -- all
eight tables compared by hash (fdcap_debug_scene_hash).  r6: 0 mismatches.   usage: python tools/stress_scene_build.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fdcap_amd
from fdcap_amd import capi, synth
ctx = capi.Context(synth.make_body_model(400, seed=0), synth.make_vposer(seed=1))
rng = np.random.default_rng(2026)
def hashes(scene, how):
    if how == "host": os.environ["FDCAP_SCENE_BUILD"] = "host"
    else: os.environ.pop("FDCAP_SCENE_BUILD", None)
    ctx.set_scene(scene)
    out = (ctypes.c_uint64 * 8)()
    capi.check(ctx.lib.fdcap_debug_scene_hash(ctx.handle, out), "hash")
    return list(out)
bad = 0
sizes = list(rng.integers(1, 3000, 150)) + list(rng.integers(3000, 60000, 60)) + [511, 512, 513, 1023, 1024, 1025, 16383, 16384, 16385, 262143, 262145, 1000003]
for k, n in enumerate(sizes):
    n = int(n)
    kind = k % 5
    if kind == 0: s = rng.uniform(-4, 4, (n, 3))
    elif kind == 1: s = rng.standard_normal((n, 3)) * [3, 0.01, 1]
    elif kind == 2: s = np.round(rng.uniform(-2, 2, (n, 3)) * 8) / 8            # heavy ties
    elif kind == 3: s = np.repeat(rng.uniform(-1, 1, (max(n // 7, 1), 3)), 7, axis=0)[:n]; 
    else: s = synth.make_scene(n, seed=int(rng.integers(1 << 30)))
    s = np.ascontiguousarray(s, dtype=np.float32)
    if s.shape[0] != n: s = np.resize(s, (n, 3)).astype(np.float32)
    a, b = hashes(s, "device"), hashes(s, "host")
    if a != b:
        bad += 1
        print("MISMATCH n", n, "kind", kind, [i for i in range(8) if a[i] != b[i]])
print("sizes tested", len(sizes), "mismatches", bad)
