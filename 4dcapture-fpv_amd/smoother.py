"""Mirror of optimization.py's FittingOP -- the older per-frame smoother -- over
`fdcap_frame_smoother` (include/fdcap.h).

Reference driver (/root/reference/optimization.py:334-348): frame 0 -> `fitting(file)`, every later frame ->
`fitting_smoothing(file, xh_prev)`; 50 Adam steps (lr 0.1) on one 78-d row per frame with
`rec + vposer (+ 5 * L1 to the previous result on columns 9:51)`; one `torch.optim.Adam` object for the
whole run, so its moments and step counter carry over between frames.  The body model, VPoser, the motion
GRU and the scene the reference's constructor loads (:106-150) take no part in these two methods.

Two ways to drive it: `fitting` / `fitting_smoothing` file by file exactly like the reference, or
`fitting_clip(rows)` -- the whole clip in one launch (same arithmetic, same order)."""
from __future__ import annotations

import glob
import os
import pickle

import numpy as np

from . import capi, io

DEFAULT_FITTINGCONFIG = {"init_lr_h": 0.1, "num_iter": 50, "batch_size": 1, "verbose": False}      # :312-318
DEFAULT_LOSSCONFIG = {"weight_loss_rec": 1, "weight_loss_vposer": 0.001, "weight_contact": 0.1,
                      "weight_collision": 0.5}                                                        # :322-327
SMOOTHING_WEIGHT = 5.0                                                                               # :227


def body_params_parse_fitting(d: dict) -> np.ndarray:
    """cvae.py:244-275: transl, global_orient, betas, body_pose, left/right_hand_pose, camera_translation."""
    return io.body_params_parse(d)


def body_params_encapsulate(xh_rec: np.ndarray) -> list:
    """cvae.py:189-208 (the 1-argument form optimization.py:284 calls)."""
    xh_rec = np.asarray(xh_rec, dtype=np.float32)
    return [{k: xh_rec[b:b + 1, lo:hi] for k, (lo, hi) in io.SLICES_75.items()} for b in range(xh_rec.shape[0])]


class FittingOP:
    def __init__(self, fittingconfig=None, lossconfig=None):
        import torch
        cfg = dict(DEFAULT_FITTINGCONFIG)
        cfg.update(fittingconfig or {})
        lcfg = dict(DEFAULT_LOSSCONFIG)
        lcfg.update(lossconfig or {})
        for k, v in list(cfg.items()) + list(lcfg.items()):
            setattr(self, k, v)
        if not torch.cuda.is_available():
            raise capi.FdcapError("no HIP device: the fdcap_amd smoother only runs on the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = capi.load_library()
        # Adam m | v | previous frame's optimised row: the state torch keeps inside self.optimizer (:126)
        self._state = torch.zeros(3, capi.XDIM, device=self.device)
        self._steps = 0

    def _rows(self, rows75):
        import torch
        x = torch.as_tensor(np.ascontiguousarray(rows75, dtype=np.float32)) if not torch.is_tensor(rows75) else rows75
        return x.to(self.device, torch.float32).reshape(-1, capi.PDIM).contiguous()

    def _run(self, rows75, has_prev):
        import torch
        x75 = self._rows(rows75)
        n = x75.shape[0]
        st = capi.current_stream()
        x78 = torch.empty(n, capi.XDIM, device=self.device)
        capi.check(self.lib.fdcap_params_75_to_78(capi.dptr(x75), n, capi.dptr(x78), st), "fdcap_params_75_to_78")   # :191
        out78 = torch.empty_like(x78)
        capi.check(self.lib.fdcap_frame_smoother(None, capi.dptr(x78), n, int(self.num_iter), float(self.init_lr_h),
                                                 float(self.weight_loss_rec), float(self.weight_loss_vposer),
                                                 SMOOTHING_WEIGHT, capi.dptr(self._state), self._steps,
                                                 1 if has_prev else 0, capi.dptr(out78), st), "fdcap_frame_smoother")
        self._steps += n * int(self.num_iter)
        out75 = torch.empty(n, capi.PDIM, device=self.device)
        capi.check(self.lib.fdcap_params_78_to_75(capi.dptr(out78), n, capi.dptr(out75), st), "fdcap_params_78_to_75")  # :206
        return out75

    @staticmethod
    def _load(input_data_file):
        with open(input_data_file, "rb") as f:
            try:
                d = pickle.load(f)
            except UnicodeDecodeError:
                f.seek(0)
                d = pickle.load(f, encoding="latin1")
        return body_params_parse_fitting(d)

    # ---- :185-208 ---------------------------------------------------------------------------
    def fitting(self, input_data_file):
        return self._run(self._load(input_data_file), has_prev=False)

    # ---- :211-238 (xh_prev is the state this object already holds; the argument is checked against it) ----
    def fitting_smoothing(self, input_data_file, xh_prev=None):
        if self._steps == 0:
            raise capi.FdcapError("fitting_smoothing needs a previous frame: call fitting() on the first file (:336-341)")
        return self._run(self._load(input_data_file), has_prev=True)

    def fitting_clip(self, rows75):
        """rows75 [N,75] (file order).  Equivalent to fitting(file 0) + fitting_smoothing(file i) for i >= 1."""
        return self._run(rows75, has_prev=self._steps > 0)

    # ---- :277-294 ---------------------------------------------------------------------------
    def save_result(self, xh_rec, output_data_file):
        import torch
        a = xh_rec.detach().cpu().numpy() if torch.is_tensor(xh_rec) else np.asarray(xh_rec)
        dirname = os.path.dirname(output_data_file)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
        for body_param in body_params_encapsulate(a):
            with open(output_data_file, "wb") as f:
                pickle.dump(body_param, f)


def main(argv=None):
    """python3 optimization.py <gen_path> <fit_path>   (optimization.py:297-349)"""
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print("usage: optimization_hip.py <gen_path> <fit_path>")
        return 2
    gen_path, fit_path = argv
    files = sorted(glob.glob(os.path.join(gen_path, "results/*/*.pkl")))                   # :330
    if not files:
        raise FileNotFoundError(f"no SMPLify-X results under {gen_path}/results/*/*.pkl")
    fop = FittingOP()
    rows = np.vstack([FittingOP._load(f) for f in files])
    out = fop.fitting_clip(rows).cpu().numpy()
    for ii in range(len(files)):
        fop.save_result(out[ii:ii + 1], os.path.join(fit_path, "smoothed_body", "{:06d}.pkl".format(ii)))   # :338
    print("[INFO][fitting] fitting finish, returning optimal value")
    return 0
