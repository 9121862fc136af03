"""BASELINE config 4 (per-frame inner fit, five stages, 64 frames batched on one GPU): Adam (round 3) against L-BFGS with a
strong-Wolfe line search (round 4, csrc/fdc_lbfgs.h) -- wall time of the whole fit, objective evaluations, the objective and
the reprojection error reached.  Synthetic case of tests/test_gpu_innerfit.py.
    python tools/innerfit_bench.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdcap_amd  # noqa: E402,F401
from fdcap_amd.innerfit import InnerFitOP  # noqa: E402
from oracle.innerfit import InnerFitOracle  # noqa: E402  (only to project joints for the printed pixel error)
from oracle.smplx import SMPLXOracle  # noqa: E402
from oracle.vposer import VPoserDecoder  # noqa: E402
from oracle import rotrepr  # noqa: E402
from tests.test_gpu_innerfit import _case  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    bm, vp, gt, init, kp = _case(n, 21)
    orc = InnerFitOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp))

    def px(rows):
        with torch.no_grad():
            uv = orc.project(orc.joints_cam(rotrepr.convert_to_6D_rot(torch.tensor(rows)))).numpy()
        w = kp[..., 2] > 0
        return float(np.sqrt(((uv - kp[..., :2]) ** 2).sum(-1))[w].mean())

    def objective(rows):
        with torch.no_grad():
            return float(sum(orc.loss(rotrepr.convert_to_6D_rot(torch.tensor(rows)), torch.tensor(kp), (1.0, 4.78, 5.0, 4.78))))
    print(f"{n} frames, five stages; start: objective {objective(init):.1f}, mean reprojection error {px(init):.2f} px")
    for name, kw in (("Adam, 30 steps per stage, lr 0.01", dict(optimizer="adam", iters_per_stage=30)),
                     ("Adam, 300 steps per stage, lr 0.01", dict(optimizer="adam", iters_per_stage=300)),
                     ("L-BFGS / strong Wolfe, SMPLify-X's settings (30 x 30, ftol 2e-9)", dict(optimizer="lbfgs")),
                     ("L-BFGS / strong Wolfe, 5 x 20 per stage", dict(optimizer="lbfgs", lbfgs=dict(max_iter=20, max_steps=5)))):
        best = None
        for rep in range(3):
            op = InnerFitOP(bm, vp, n, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = op.fitting(init, kp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            rounds = sum(op.rounds) if op.rounds else 5 * kw["iters_per_stage"]
            worst = ""
            if op.frame_loss:                                  # the three frames that ended highest, and how many directions they took
                fl = op.frame_loss[-1]
                top = np.argsort(-fl)[:3]
                worst = "; highest frames " + ", ".join(f"#{i}: {fl[i]:.0f} ({sum(int(it[i]) for it in op.frame_iterations)} directions)" for i in top)
            op.close()
        out = out.cpu().numpy()
        print(f"{name:70s}: {best * 1e3:8.1f} ms = {n / best:7.1f} frames/s, {rounds:5d} objective evaluations ({best / rounds * 1e6:6.1f} us each), "
              f"objective {objective(out):.1f}, reprojection error {px(out):.2f} px{worst}")


if __name__ == "__main__":
    main()
