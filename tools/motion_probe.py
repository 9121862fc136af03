#!/usr/bin/env python3
"""How fast do the Chamfer queries (world contact vertices) move from one optimiser iteration to the next, and how
long would a cached per-group work list of the in-loop NN launch stay valid?  (Design input for the list cache of
fdc::nn_stream4_kernel; analysis only -- torch is used for the statistics, the fit itself is the HIP path.)

  python tools/motion_probe.py [--frames 1024] [--scene 500000] [--iters 500] > gpurun_out/motion_probe.json
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def morton_order(vt, vid):
    p = vt[vid]
    lo, hi = p.min(0), p.max(0)
    u = np.where(hi > lo, (p - lo) / np.where(hi > lo, hi - lo, 1), 0.0)
    q = np.clip((u * 1023.0), 0, 1023).astype(np.uint32)

    def spread(v):
        v = v & 1023
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.lexsort((np.arange(len(vid)), code))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--scene", type=int, default=500_000)
    ap.add_argument("--iters", type=int, default=500)
    args = ap.parse_args()
    import fdcap_amd  # noqa: F401
    from fdcap_amd import capi, synth
    from fdcap_amd.fitting import FittingOP, first_phase2_iter
    from fdcap_amd.io import read_camerapose
    N = args.frames
    bm = synth.make_body_model(10475, seed=0)
    vp = synth.make_vposer(seed=1)
    clip = synth.make_clip(N, seed=3)
    scene = synth.make_scene(args.scene, seed=2)
    left, right = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
    vid = np.concatenate([left, right])
    nc = len(vid)
    fop = FittingOP({"num_iter": args.iters}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines))
    lib, h = fop.ctx.lib, fop.ctx.handle
    dev = fop.device
    body = torch.tensor(clip.body_params).cuda()
    x78 = torch.empty(N, capi.XDIM, device=dev)
    capi.check(lib.fdcap_params_75_to_78(capi.dptr(body), N, capi.dptr(x78), capi.current_stream()), "75->78")
    fop._mode = "global"
    fop.init(x78)
    P = first_phase2_iter(args.iters)
    order = torch.tensor(morton_order(bm.v_template, vid).copy(), device=dev)      # internal slot -> caller position
    sc = torch.tensor(scene, device=dev)
    verts = torch.empty(N, nc, 3, device=dev)
    dist = torch.empty(N, nc, device=dev)
    idx = torch.empty(N, nc, device=dev, dtype=torch.int32)
    slacks = [0.005, 0.01, 0.02, 0.04]
    G = (N * nc) // 32
    anchors = {s: None for s in slacks}
    rows = []
    prev = None
    prev_nn = None
    for ii in range(P):
        st = capi.current_stream()
        capi.check(lib.fdcap_opt_forward_world(h, capi.dptr(verts), None, st), "forward_world")
        capi.check(lib.fdcap_opt_get_contact(h, capi.dptr(dist), capi.dptr(idx), st), "get_contact")
        x = verts[:, order, :].reshape(-1, 3)[: G * 32]                               # internal (launch) order
        d1 = dist[:, order].reshape(-1)[: G * 32].clamp_min(0).sqrt()
        nn = sc[idx[:, order].reshape(-1)[: G * 32].long()]
        row = {"ii": ii}
        qs = torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)
        row["d1_q"] = torch.quantile(d1[::7], qs).tolist()
        if prev is not None:
            delta = (x - prev).norm(dim=1)
            row["delta_q"] = torch.quantile(delta[::7], qs).tolist()
            bound = (x - prev_nn).norm(dim=1)                                         # the seed's distance = the scan's initial bound
            row["seed_excess_q"] = torch.quantile((bound - d1)[::7], qs).tolist()
            for s in slacks:
                a = anchors[s]
                if a is None:
                    anchors[s] = [x.clone(), bound + s]
                    row[f"rebuild_{s}"] = 1.0
                    continue
                ok = (bound + (x - a[0]).norm(dim=1)) <= a[1]
                gok = ok.view(G, 32).all(dim=1)
                row[f"rebuild_{s}"] = float((~gok).float().mean())
                reb = (~gok).repeat_interleave(32)
                a[0][reb] = x[reb]
                a[1][reb] = bound[reb] + s
        prev, prev_nn = x.clone(), nn.clone()
        rows.append(row)
        capi.check(lib.fdcap_opt_backward(h, ii, P, 0, st), "backward")
        capi.check(lib.fdcap_opt_step(h, ii, P, st), "step")
    torch.cuda.synchronize()
    summ = {}
    for s in slacks:
        r = np.array([row.get(f"rebuild_{s}", 1.0) for row in rows[2:]])
        summ[str(s)] = {"mean_rebuild": float(r.mean()), "first100": float(r[:100].mean()), "last100": float(r[-100:].mean())}
    print(json.dumps({"summary": summ, "rows": rows[:5] + rows[5::25]}, indent=1))


if __name__ == "__main__":
    main()
