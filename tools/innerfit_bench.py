"""BASELINE config 4 (per-frame inner fit, five stages, 64 frames batched on one GPU): Adam (round 3) against L-BFGS with a
strong-Wolfe line search (round 4, csrc/fdc_lbfgs.h) -- wall time of the whole fit, objective evaluations, the objective and
the reprojection error reached.  The synthetic case of tests/test_gpu_innerfit.py, built here with the library's own forward
(ground-truth rows -> camera-frame joints -> pinhole projection + 2 px noise + confidences; start = perturbed ground truth).
    python tools/innerfit_bench.py [frames]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdcap_amd  # noqa: E402,F401
from fdcap_amd import capi, synth  # noqa: E402
from fdcap_amd.innerfit import DEFAULT_INTRINSICS, DEFAULT_STAGES, InnerFitOP  # noqa: E402


def joints_cam(op, n):
    """Camera-frame joints 0..22 of the optimiser's current rows, by the library's forward."""
    j = torch.empty(n, 23, 3, device="cuda")
    capi.check(op.ctx.lib.fdcap_opt_forward_world(op.ctx.handle, None, capi.dptr(j), capi.current_stream()), "fdcap_opt_forward_world")
    return j.cpu().numpy()


def project(j):
    fx, fy, cx, cy = DEFAULT_INTRINSICS
    return np.stack([fx * j[..., 0] / j[..., 2] + cx, fy * j[..., 1] / j[..., 2] + cy], -1)


def make_case(n, seed):
    bm = synth.make_body_model(240, seed=seed)
    vp = synth.make_vposer(seed=seed + 1)
    clip = synth.make_clip(n, seed=seed + 2, num_outliers=1)
    rng = np.random.Generator(np.random.PCG64(seed + 3))
    gt = clip.body_params.astype(np.float32).copy()
    gt[:, 72:75] = np.array([0.1, -0.2, 3.5], np.float32) + 0.05 * rng.standard_normal((n, 3)).astype(np.float32)
    op = InnerFitOP(bm, vp, n, iters_per_stage=0)
    op.fitting(gt, np.zeros((n, 23, 3), np.float32))                # zero iterations: the state is the ground truth
    uv = project(joints_cam(op, n))
    op.close()
    kp = np.concatenate([uv + 2.0 * rng.standard_normal(uv.shape), rng.uniform(0.3, 1.0, (n, 23, 1))], -1).astype(np.float32)
    kp[:, 22, 2] = 0.0
    init = gt.copy()
    init[:, 3:6] += 0.15 * rng.standard_normal((n, 3)).astype(np.float32)
    init[:, 16:48] += 0.5 * rng.standard_normal((n, 32)).astype(np.float32)
    init[:, 72:75] += 0.1 * rng.standard_normal((n, 3)).astype(np.float32)
    return bm, vp, init, kp


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    bm, vp, init, kp = make_case(n, 21)
    w = kp[..., 2] > 0
    last = DEFAULT_STAGES[-1]

    def state_of(op):
        """objective at the last stage's weights and mean reprojection error (px) of the optimiser's current rows"""
        sg = capi.Fit2dStage(*DEFAULT_INTRINSICS, 100.0, *last)
        capi.check(op.ctx.lib.fdcap_opt_backward_fit2d(op.ctx.handle, ctypes.byref(sg), 1, capi.current_stream()), "fdcap_opt_backward_fit2d")
        s = op._losses.cpu().numpy()
        px = float(np.sqrt(((project(joints_cam(op, n)) - kp[..., :2]) ** 2).sum(-1))[w].mean())
        return float(s[0] + s[1]), px
    op = InnerFitOP(bm, vp, n, iters_per_stage=0)
    op.fitting(init, kp)
    o0, p0 = state_of(op)
    op.close()
    print(f"{n} frames, five stages; start: objective {o0:.1f}, mean reprojection error {p0:.2f} px")
    for name, kw in (("Adam, 30 steps per stage, lr 0.01", dict(optimizer="adam", iters_per_stage=30)),
                     ("Adam, 300 steps per stage, lr 0.01", dict(optimizer="adam", iters_per_stage=300)),
                     ("L-BFGS / strong Wolfe, SMPLify-X's settings (30 x 30, ftol 2e-9)", dict(optimizer="lbfgs")),
                     ("L-BFGS / strong Wolfe, 5 x 20 per stage", dict(optimizer="lbfgs", lbfgs=dict(max_iter=20, max_steps=5)))):
        best = None
        for rep in range(3):
            op = InnerFitOP(bm, vp, n, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            op.fitting(init, kp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            rounds = sum(op.rounds) if op.rounds else 5 * kw["iters_per_stage"]
            worst = ""
            if op.frame_loss:                                  # the three frames that ended highest, and how many directions they took
                fl = op.frame_loss[-1]
                top = np.argsort(-fl)[:3]
                worst = "; highest frames " + ", ".join(f"#{i}: {fl[i]:.0f} ({sum(int(it[i]) for it in op.frame_iterations)} directions)" for i in top)
            obj, px = state_of(op)
            op.close()
        print(f"{name:70s}: {best * 1e3:8.1f} ms = {n / best:7.1f} frames/s, {rounds:5d} objective evaluations ({best / rounds * 1e6:6.1f} us each), "
              f"objective {obj:.1f}, reprojection error {px:.2f} px{worst}")


if __name__ == "__main__":
    main()
