"""Restatement of optimization.py's per-frame smoother.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/optimization.py:
  cal_loss :155-163 · smoothing_loss :173-183 · fitting :185-208 · fitting_smoothing :211-238 ·
  the driver loop :334-348 (frame 0 -> fitting, frame i -> fitting_smoothing(file, previous result)).
One torch.optim.Adam for the whole run (:126): its moments and step counter carry over between frames.
Pinned by tests/golden/ref_smoother.npz, which the reference's own code produced
(tests/golden/make_golden.py --smoother)."""
import torch
import torch.nn.functional as F

from . import rotrepr


class SmootherOracle:
    def __init__(self, init_lr_h=0.1, num_iter=50, weight_loss_rec=1.0, weight_loss_vposer=0.001, dtype=torch.float32):
        self.num_iter = num_iter
        self.weight_loss_rec = weight_loss_rec
        self.weight_loss_vposer = weight_loss_vposer
        self.dtype = dtype
        self.xhr_rec = torch.zeros(1, 78, dtype=dtype, requires_grad=True)           # :125 (+ the .data swap of :192)
        self.optimizer = torch.optim.Adam([self.xhr_rec], lr=init_lr_h)                # :126

    def cal_loss(self, xhr):
        loss_rec = self.weight_loss_rec * F.l1_loss(xhr, self.xhr_rec)                 # :157
        xh_rec = rotrepr.convert_to_3D_rot(self.xhr_rec)
        loss_vposer = self.weight_loss_vposer * torch.mean(xh_rec[:, 16:48] ** 2)      # :161-162
        return loss_rec, loss_vposer

    def fitting(self, xh75, xh_prev75=None):
        xhr = rotrepr.convert_to_6D_rot(torch.as_tensor(xh75).to(self.dtype).reshape(1, 75))
        prev = None if xh_prev75 is None else rotrepr.convert_to_6D_rot(xh_prev75)     # :219
        self.xhr_rec.data = xhr.clone()
        for _ in range(self.num_iter):
            self.optimizer.zero_grad()
            loss_rec, loss_vposer = self.cal_loss(xhr)
            loss = loss_rec + loss_vposer
            if prev is not None:
                loss = loss + F.l1_loss(prev[:, 9:51], self.xhr_rec[:, 9:51]) * 5      # :182, :227
            loss.backward()
            self.optimizer.step()
        return rotrepr.convert_to_3D_rot(self.xhr_rec).detach()

    def fitting_clip(self, rows75):
        """The driver loop :334-348 over [N,75] rows -> [N,75]."""
        out, prev = [], None
        for r in torch.as_tensor(rows75).to(self.dtype):
            prev = self.fitting(r, prev)
            out.append(prev)
        return torch.cat(out, dim=0)
