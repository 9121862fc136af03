#!/usr/bin/env python3
"""Why are the searches right after the phase switch slow (profiles/r4_nn_logging_series.txt), and would a better seed help?
Every-iteration-logging fit on the bench workload; at chosen iterations: the radius the search starts from (distance from a
query's new position to ITS neighbour of the iteration before), the radius it would start from with the best of the previous
neighbours of the 32 contact vertices nearest to it on the body (what a group-wide seed could offer), and the true distance."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd  # noqa
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N = 1024
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500_000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
ids = np.concatenate([l, r]); nc = len(ids)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=ids,
                camera_ext=read_camerapose(clip.camerapose_lines))
S = torch.tensor(scene, device="cuda")
rest = torch.tensor(bm.v_template[ids], device="cuda")
nbr = torch.cdist(rest, rest).topk(32, largest=False).indices            # [nc,32] the 32 nearest contact vertices on the body
ks = [100, 101, 300, 301, 399, 400, 401, 402, 403, 405, 408, 412, 420, 440, 480]
rec = {}
def hook(k):
    lib, h, st = fop.ctx.lib, fop.ctx.handle, capi.current_stream()
    d = torch.empty(N, nc, device="cuda"); idx = torch.empty(N, nc, dtype=torch.int32, device="cuda")
    capi.check(lib.fdcap_opt_get_contact(h, capi.dptr(d), capi.dptr(idx), st), "get_contact")     # search of iteration k-1 (positions before step k)
    v = torch.empty(N, nc, 3, device="cuda")
    capi.check(lib.fdcap_opt_forward_world(h, capi.dptr(v), None, st), "forward_world")             # positions after step k = queries of iteration k
    torch.cuda.synchronize()
    rec[k] = (d.sqrt().clone(), idx.long().clone(), v.clone())
fop.snapshot_hook = hook
fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1, snapshot_at=ks + [k + 1 for k in ks])
print("iteration: median / 90 % of [true distance | start radius from the own previous neighbour | from the best previous neighbour of the 32 nearest contact vertices]  (cm)")
for k in ks:
    if k not in rec or k + 1 not in rec: continue
    d_prev, idx_prev, v = rec[k]
    d_true = rec[k + 1][0]                                           # the search of iteration k
    own = (v - S[idx_prev]).norm(dim=-1)
    best = torch.empty_like(own)
    for f0 in range(0, N, 64):
        P = S[idx_prev[f0:f0 + 64]]                                  # [f,nc,3]
        cand = P[:, nbr]                                             # [f,nc,32,3]
        best[f0:f0 + 64] = (v[f0:f0 + 64, :, None, :] - cand).norm(dim=-1).min(dim=-1).values
    q = lambda t: f"{t.median()*100:.2f} / {t.flatten().kthvalue(int(0.9*t.numel())).values*100:.2f}"
    vol = ((own / d_true.clamp_min(1e-4)) ** 3).median(), ((best / d_true.clamp_min(1e-4)) ** 3).median()
    print(f"{k:4d}: {q(d_true)} | {q(own)} | {q(best)}   median (radius / true)^3: own {vol[0]:.1f}, group {vol[1]:.1f}")
