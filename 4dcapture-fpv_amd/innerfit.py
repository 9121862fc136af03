"""Per-frame inner fit with a true 2D-keypoint reprojection residual, batched over frames on one GPU
(BASELINE config 4) over `fdcap_opt_backward_fit2d` (include/fdcap.h, csrc/fdc_fit2d.h).

Not part of the reference repository (there it is the external SMPLify-X step, README.md:14-17); the objective is the
published SMPLify-X data term + L2 priors in five weight stages, optimised either as SMPLify-X does -- L-BFGS with a
strong-Wolfe line search, every frame its own problem, a fresh optimiser per stage (`optimizer="lbfgs"`: the batched state
machine of csrc/fdc_lbfgs.h through `fdcap_opt_fit2d_lbfgs`) -- or with Adam (`optimizer="adam"`, round 3's form).
Inputs / outputs use the repository's own parameter layout ([N,75] rows, SMPLify-X pkl keys), so the result can feed
`global_optimization_hip.py` directly."""
from __future__ import annotations

import ctypes

import numpy as np

from . import capi

# (w_data, w_pose, w_shape, w_hand) per stage
DEFAULT_STAGES = ((1.0, 404.0, 100.0, 404.0), (1.0, 404.0, 50.0, 404.0), (1.0, 57.4, 10.0, 57.4), (1.0, 4.78, 5.0, 4.78),
                  (1.0, 4.78, 5.0, 4.78))
DEFAULT_INTRINSICS = (692.0, 692.0, 640.0, 360.0)      # vis.py:358-360
# SMPLify-X's optimiser settings (fit_smplx.yaml: maxiters 30, ftol 2e-9, gtol 1e-9, lr 1.0) over torch.optim.LBFGS's
# defaults (history 100, tolerance_grad 1e-7, tolerance_change 1e-9, max_eval = max_iter * 5 / 4, 25 line-search evaluations)
DEFAULT_LBFGS = dict(history=100, max_iter=30, max_eval=0, max_steps=30, max_ls=25, lr=1.0, tolerance_grad=1e-7,
                     tolerance_change=1e-9, ftol=2e-9, gtol=1e-9)


class InnerFitOP:
    def __init__(self, body_model, vposer, num_frames, intrinsics=DEFAULT_INTRINSICS, rho=100.0, lr=0.01,
                 stages=DEFAULT_STAGES, iters_per_stage=30, optimizer="adam", lbfgs=None, max_rounds=4000):
        """optimizer "adam": iters_per_stage steps of Adam(lr) per stage; "lbfgs": per frame and stage, SMPLify-X's L-BFGS loop
        with the settings `lbfgs` (keys of DEFAULT_LBFGS; missing ones take their defaults), at most max_rounds objective
        evaluations per stage."""
        import torch
        if not torch.cuda.is_available():
            raise capi.FdcapError("no HIP device: the fdcap_amd inner fit only runs on the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = capi.Context(body_model, vposer)
        self.ctx.set_scene(np.zeros((0, 3), np.float32))
        self.n = int(num_frames)
        self.intrinsics, self.rho, self.lr = tuple(float(v) for v in intrinsics), float(rho), float(lr)
        self.stages, self.iters_per_stage = tuple(stages), int(iters_per_stage)
        if optimizer not in ("adam", "lbfgs"):
            raise capi.FdcapError(f"optimizer must be 'adam' or 'lbfgs', not {optimizer!r}")
        unknown = set(lbfgs or {}) - set(DEFAULT_LBFGS)
        if unknown:
            raise capi.FdcapError(f"unknown L-BFGS settings {sorted(unknown)}")
        self.optimizer, self.lbfgs, self.max_rounds = optimizer, {**DEFAULT_LBFGS, **(lbfgs or {})}, int(max_rounds)
        self.log, self.rounds, self.frame_iterations, self.frame_loss = [], [], [], []

    def fitting(self, rows75, keypoints, log_every=0):
        """rows75 [N,75] initial parameters (device tensor or numpy), keypoints [N,23,3] (u, v, confidence)
        -> optimised rows [N,75] (device tensor)."""
        import torch
        lib, h, dev, n = self.ctx.lib, self.ctx.handle, self.device, self.n
        t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev, torch.float32).contiguous()
        rows75, kp = t(rows75), t(keypoints)
        if tuple(rows75.shape) != (n, capi.PDIM) or tuple(kp.shape) != (n, 23, 3):
            raise capi.FdcapError(f"expected rows [{n},75] and keypoints [{n},23,3]")
        st = capi.current_stream()
        x78 = torch.empty(n, capi.XDIM, device=dev)
        capi.check(lib.fdcap_params_75_to_78(capi.dptr(rows75), n, capi.dptr(x78), st), "fdcap_params_75_to_78")
        # the optimiser state of the clip-level loop, with scale = 1 and camera_ext = identity: its "world" joints are
        # then the camera-frame joints + camera_translation; no contact / temporal term is ever evaluated
        oc = capi.OptConfig(n, n, 0, self.lr, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0)
        R = n + 4
        self._rows_x = torch.zeros(R, capi.XDIM, device=dev)
        self._rows_cam = torch.zeros(R, 16, device=dev)
        self._scale, self._dscale = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        self._losses = torch.zeros(capi.NUM_LOSSES, device=dev, dtype=torch.float64)
        torch.cuda.current_stream().synchronize()
        capi.check(lib.fdcap_opt_create(h, ctypes.byref(oc), capi.dptr(self._rows_x), capi.dptr(self._rows_cam),
                                        capi.dptr(self._scale), capi.dptr(self._dscale), capi.dptr(self._losses)), "fdcap_opt_create")
        eye = torch.eye(4, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
        self._rows_cam[:] = torch.eye(4, device=dev).reshape(1, 16)          # halo rows too (their forward is evaluated, never used)
        capi.check(lib.fdcap_opt_set_inputs(h, capi.dptr(x78), capi.dptr(x78), capi.dptr(torch.ones(n, device=dev)), capi.dptr(eye), st),
                   "fdcap_opt_set_inputs")
        capi.check(lib.fdcap_opt_set_keypoints(h, capi.dptr(kp), st), "fdcap_opt_set_keypoints")
        self.log, self.rounds, self.frame_iterations, self.frame_loss = [], [], [], []
        fx, fy, cx, cy = self.intrinsics
        for w_data, w_pose, w_shape, w_hand in self.stages:
            sg = capi.Fit2dStage(fx, fy, cx, cy, self.rho, w_data, w_pose, w_shape, w_hand)
            if self.optimizer == "lbfgs":
                q = self.lbfgs
                cf = capi.LbfgsConfig(capi.XDIM, q["history"], q["max_iter"], q["max_eval"], q["max_steps"], q["max_ls"], q["lr"],
                                      q["tolerance_grad"], q["tolerance_change"], q["ftol"], q["gtol"])
                rounds = ctypes.c_int32(0)
                capi.check(lib.fdcap_opt_fit2d_lbfgs(h, ctypes.byref(sg), ctypes.byref(cf), self.max_rounds, ctypes.byref(rounds),
                                                     capi.current_stream()), "fdcap_opt_fit2d_lbfgs")
                self.rounds.append(int(rounds.value))
                it = torch.zeros(n, dtype=torch.int32, device=dev)
                fl = torch.zeros(n, device=dev)
                capi.check(lib.fdcap_opt_fit2d_lbfgs_stats(h, capi.dptr(it), None, capi.dptr(fl), capi.current_stream()),
                           "fdcap_opt_fit2d_lbfgs_stats")
                self.frame_iterations.append(it.cpu().numpy())      # per frame: L-BFGS directions taken in this stage
                self.frame_loss.append(fl.cpu().numpy())            # ... and the objective where it stopped
                if log_every:                                   # the objective where the stage ended
                    capi.check(lib.fdcap_opt_backward_fit2d(h, ctypes.byref(sg), 1, capi.current_stream()), "fdcap_opt_backward_fit2d")
                    s = self._losses.cpu().numpy()
                    self.log.append([float(s[0]), float(s[1])])
                continue
            capi.check(lib.fdcap_opt_reset_adam(h, capi.current_stream()), "fdcap_opt_reset_adam")
            for it in range(self.iters_per_stage):
                st = capi.current_stream()
                do_log = bool(log_every) and it % log_every == 0
                capi.check(lib.fdcap_opt_backward_fit2d(h, ctypes.byref(sg), 1 if do_log else 0, st), "fdcap_opt_backward_fit2d")
                if do_log:
                    s = self._losses.cpu().numpy()
                    self.log.append([float(s[0]), float(s[1])])
                capi.check(lib.fdcap_opt_step_x(h, it + 1, st), "fdcap_opt_step_x")
        out = torch.empty(n, capi.PDIM, device=dev)
        capi.check(lib.fdcap_opt_get_results(h, capi.dptr(out), None, None, capi.current_stream()), "fdcap_opt_get_results")
        self.body_rotation_rec = self._rows_x[2:2 + n]
        return out

    def close(self):
        if getattr(self, "ctx", None) is not None:
            self.ctx.close()
            self.ctx = None
