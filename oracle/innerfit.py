"""Per-frame inner fit with a 2D-keypoint reprojection term (BASELINE config 4).

TEST INFRASTRUCTURE (see oracle/__init__.py).  NOT restated from /root/reference: the per-frame fit is the external
SMPLify-X step there (README.md:14-17).  **Parity unpinned**: the objective is written down from the published SMPLify-X
data term (GMoF-robust, confidence-weighted joint reprojection through a pinhole camera; the in-repo viewers use
fx = fy = 692, cx = 640, cy = 360, vis.py:358-360) and its L2 priors on the VPoser latent, betas and hand PCA
coefficients, in five weight stages; the optimiser is Adam with a fresh state per stage (`fitting`, round 3) or, as in
SMPLify-X, L-BFGS with a strong-Wolfe line search (`fitting_lbfgs`: torch.optim.LBFGS ITSELF, one instance per frame and
stage, inside SMPLify-X's loop around optimizer.step() -- the checker of csrc/fdc_lbfgs.h).  This file is the autograd twin
of csrc/fdc_fit2d.h."""
import torch

from . import rotrepr
from .fitting import body_params_encapsulate_batch

# (w_data, w_pose, w_shape, w_hand) per stage: SMPLify-X's default body_pose_prior / shape / hand_prior weight schedules
DEFAULT_STAGES = ((1.0, 404.0, 100.0, 404.0), (1.0, 404.0, 50.0, 404.0), (1.0, 57.4, 10.0, 57.4), (1.0, 4.78, 5.0, 4.78),
                  (1.0, 4.78, 5.0, 4.78))
DEFAULT_INTRINSICS = (692.0, 692.0, 640.0, 360.0)


class InnerFitOracle:
    def __init__(self, body_model, vposer, intrinsics=DEFAULT_INTRINSICS, rho=100.0, lr=0.01, dtype=torch.float32):
        self.body_mesh_model, self.vposer = body_model, vposer
        self.fx, self.fy, self.cx, self.cy = intrinsics
        self.rho, self.lr, self.dtype = rho, lr, dtype

    def joints_cam(self, x78):
        """Camera-frame joints 0..22: SMPL-X joints (+transl) + camera_translation (camera rotation = identity)."""
        n = x78.shape[0]
        p = body_params_encapsulate_batch(rotrepr.convert_to_3D_rot(x78))
        joint_rot = self.vposer.decode(p["body_pose_vp"], output_type="aa").view(n, -1)
        out = self.body_mesh_model(return_verts=True, body_pose=joint_rot, transl=p["transl"], global_orient=p["global_orient"],
                                   betas=p["betas"], left_hand_pose=p["left_hand_pose"], right_hand_pose=p["right_hand_pose"])
        return out.joints[:, 0:23, :] + x78[:, 75:78].unsqueeze(1)

    def project(self, J):
        return torch.stack([self.fx * J[..., 0] / J[..., 2] + self.cx, self.fy * J[..., 1] / J[..., 2] + self.cy], dim=-1)

    def loss(self, x78, kp, stage):
        w_data, w_pose, w_shape, w_hand = stage
        res = kp[..., :2] - self.project(self.joints_cam(x78))
        r2 = res ** 2
        gm = self.rho ** 2 * r2 / (r2 + self.rho ** 2)
        data = w_data ** 2 * torch.sum(kp[..., 2:3] ** 2 * gm)
        prior = (w_pose ** 2 * torch.sum(x78[:, 19:51] ** 2) + w_shape ** 2 * torch.sum(x78[:, 9:19] ** 2)
                 + w_hand ** 2 * torch.sum(x78[:, 51:75] ** 2))
        return data, prior

    def fitting(self, rows75, keypoints, stages=DEFAULT_STAGES, iters_per_stage=30):
        """rows75 [N,75] initial per-frame parameters, keypoints [N,23,3] (u, v, conf) -> rows75 [N,75]."""
        x = rotrepr.convert_to_6D_rot(torch.as_tensor(rows75).to(self.dtype)).detach().clone().requires_grad_(True)
        kp = torch.as_tensor(keypoints).to(self.dtype)
        self.loss_log = []
        for stage in stages:
            opt = torch.optim.Adam([x], lr=self.lr)
            for _ in range(iters_per_stage):
                opt.zero_grad()
                data, prior = self.loss(x, kp, stage)
                self.loss_log.append([float(data.detach()), float(prior.detach())])
                (data + prior).backward()
                opt.step()
        self.x78 = x.detach()
        return rotrepr.convert_to_3D_rot(x).detach()

    def fitting_lbfgs(self, rows75, keypoints, stages=DEFAULT_STAGES, history=100, max_iter=30, max_eval=0, max_steps=30, lr=1.0,
                      tolerance_grad=1e-7, tolerance_change=1e-9, ftol=2e-9, gtol=1e-9):
        """Every frame its own problem, as SMPLify-X fits it: per stage a fresh torch.optim.LBFGS(strong_wolfe) and
        [upstream-recall: smplifyx FittingMonitor.run_fitting] `for n in range(maxiters): loss = optimizer.step(closure)`, stopping
        when the loss is not finite, when n > 0 and |prev - loss| / max(|prev|, |loss|, 1) <= ftol, or when every gradient entry is
        below gtol.  (gtol is tested on the ACCEPTED point's gradient here and in the kernel; SMPLify-X reads param.grad, which
        holds the last line-search trial's.)  -> rows75 [N,75]; self.evals [stage][frame] = closure calls."""
        x_all = rotrepr.convert_to_6D_rot(torch.as_tensor(rows75).to(self.dtype)).detach().clone()
        kp_all = torch.as_tensor(keypoints).to(self.dtype)
        self.evals, self.final_loss = [], []
        for stage in stages:
            ev, fl = [], []
            for f in range(x_all.shape[0]):
                x = x_all[f:f + 1].clone().requires_grad_(True)
                kp = kp_all[f:f + 1]
                opt = torch.optim.LBFGS([x], lr=lr, max_iter=max_iter, max_eval=max_eval if max_eval > 0 else None,
                                        history_size=history, tolerance_grad=tolerance_grad, tolerance_change=tolerance_change,
                                        line_search_fn="strong_wolfe")
                calls = [0]

                def closure():
                    opt.zero_grad()
                    data, prior = self.loss(x, kp, stage)
                    loss = data + prior
                    loss.backward()
                    calls[0] += 1
                    return loss
                prev = None
                for n in range(max_steps):
                    loss = float(opt.step(closure).detach())
                    if not (abs(loss) < float("inf")):
                        break
                    if n > 0 and prev is not None and ftol > 0 and abs(prev - loss) / max(abs(prev), abs(loss), 1.0) <= ftol:
                        break
                    with torch.enable_grad():
                        g = torch.autograd.grad(sum(self.loss(x, kp, stage)), x)[0]
                    if float(g.abs().max()) < gtol:
                        break
                    prev = loss
                x_all[f] = x.detach()[0]
                ev.append(calls[0])
                with torch.no_grad():
                    fl.append(float(sum(self.loss(x, kp, stage))))
            self.evals.append(ev)
            self.final_loss.append(fl)
        self.x78 = x_all
        return rotrepr.convert_to_3D_rot(x_all).detach()
