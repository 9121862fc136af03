"""GPU tests of the per-frame inner fit with a 2D-keypoint reprojection term (BASELINE config 4:
64 frames batched on one GPU, five weight stages) through the C-ABI, against the oracle's autograd twin
(oracle/innerfit.py).  The objective is outside the reference repository: parity here is HIP path vs oracle only."""
import ctypes

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi, synth
from fdcap_amd.innerfit import DEFAULT_STAGES, InnerFitOP
from oracle import rotrepr
from oracle.innerfit import InnerFitOracle
from oracle.smplx import SMPLXOracle
from oracle.vposer import VPoserDecoder

pytestmark = pytest.mark.gpu


def _case(n, seed):
    """Ground-truth clip -> projected joints (+ pixel noise, confidences) as keypoints; start = perturbed ground truth."""
    bm = synth.make_body_model(240, seed=seed)
    vp = synth.make_vposer(seed=seed + 1)
    clip = synth.make_clip(n, seed=seed + 2, num_outliers=1)
    rng = np.random.Generator(np.random.PCG64(seed + 3))
    gt = clip.body_params.astype(np.float32).copy()
    gt[:, 72:75] = np.array([0.1, -0.2, 3.5], np.float32) + 0.05 * rng.standard_normal((n, 3)).astype(np.float32)
    orc = InnerFitOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp))
    with torch.no_grad():
        uv = orc.project(orc.joints_cam(rotrepr.convert_to_6D_rot(torch.tensor(gt)))).numpy()
    kp = np.concatenate([uv + 2.0 * rng.standard_normal(uv.shape), rng.uniform(0.3, 1.0, (n, 23, 1))], -1).astype(np.float32)
    kp[:, 22, 2] = 0.0                                              # one joint never detected
    init = gt.copy()
    init[:, 3:6] += 0.15 * rng.standard_normal((n, 3)).astype(np.float32)
    init[:, 16:48] += 0.5 * rng.standard_normal((n, 32)).astype(np.float32)
    init[:, 72:75] += 0.1 * rng.standard_normal((n, 3)).astype(np.float32)
    return bm, vp, gt, init, kp


def test_fit2d_gradient_matches_autograd():
    n = 16
    bm, vp, gt, init, kp = _case(n, 7)
    op = InnerFitOP(bm, vp, n, iters_per_stage=0)
    op.fitting(init, kp)                                           # sets everything up, zero iterations
    lib, h = op.ctx.lib, op.ctx.handle
    sg = capi.Fit2dStage(692, 692, 640, 360, 100.0, 1.0, 4.78, 5.0, 4.78)
    capi.check(lib.fdcap_opt_backward_fit2d(h, ctypes.byref(sg), 1, capi.current_stream()), "backward_fit2d")
    dx = torch.empty(n, 78, device="cuda")
    capi.check(lib.fdcap_opt_get_grads(h, capi.dptr(dx), None, capi.current_stream()), "get_grads")
    s = op._losses.cpu().numpy()
    orc = InnerFitOracle(SMPLXOracle(bm, dtype=torch.float64), VPoserDecoder.from_data(vp, dtype=torch.float64), dtype=torch.float64)
    x = op.body_rotation_rec.detach().cpu().double().requires_grad_(True)
    data, prior = orc.loss(x, torch.tensor(kp, dtype=torch.float64), (1.0, 4.78, 5.0, 4.78))
    (data + prior).backward()
    np.testing.assert_allclose(s[:2], [float(data), float(prior)], rtol=2e-5)
    g = x.grad.numpy()
    np.testing.assert_allclose(dx.cpu().numpy(), g, rtol=2e-3, atol=2e-4 * np.abs(g).max())
    op.close()


def test_config4_five_stage_fit_matches_oracle_and_reduces_the_reprojection_error():
    """BASELINE config 4: 64 frames batched on one GPU, five stages (SMPLify-X's prior-weight schedule)."""
    n, iters = 64, 20
    bm, vp, gt, init, kp = _case(n, 21)
    op = InnerFitOP(bm, vp, n, iters_per_stage=iters)
    out = op.fitting(init, kp, log_every=1).cpu().numpy()
    orc = InnerFitOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp))
    ref = orc.fitting(init, kp, DEFAULT_STAGES, iters).numpy()
    # GMoF + L2 are smooth, but the latent's gradient runs through VPoser's LeakyReLUs: a pre-activation that changes sign
    # between two fp32 evaluations changes a slope, and Adam (lr 0.01, 100 steps) amplifies it.  Measured: q99 8e-4, max
    # 0.036 (latent; betas 0.015, rotation 8e-3, translations 1.6e-3, hands 5e-8)
    err = np.abs(out - ref)
    print("inner fit vs oracle: max", err.max(), "q99", np.quantile(err, 0.99), "per block (transl, aa, betas, latent, hands, cam_t)",
          [float(err[:, a:b].max()) for a, b in ((0, 3), (3, 6), (6, 16), (16, 48), (48, 72), (72, 75))])
    assert np.quantile(err, 0.5) < 2e-5 and np.quantile(err, 0.99) < 3e-3 and err.max() < 0.1 and err[:, 48:72].max() < 1e-6
    log, olog = np.array(op.log), np.array(orc.loss_log)
    np.testing.assert_allclose(log[:5], olog[:5], rtol=1e-4)                # before any drift: the same objective values
    np.testing.assert_allclose(log, olog, rtol=2e-2, atol=1e-3)

    def px_err(rows):
        with torch.no_grad():
            uv = orc.project(orc.joints_cam(rotrepr.convert_to_6D_rot(torch.tensor(rows)))).numpy()
        w = kp[..., 2] > 0
        return float(np.sqrt(((uv - kp[..., :2]) ** 2).sum(-1))[w].mean())
    e0, e1 = px_err(init), px_err(out)
    print("mean reprojection error (px): start", e0, "end", e1)
    assert e1 < 0.5 * e0
    op.close()


def test_undetected_keypoints_leave_only_the_priors():
    """confidence 0 everywhere: the data term and its gradient vanish, the L2 priors shrink latent / betas / hands and
    nothing else moves."""
    n = 8
    bm, vp, gt, init, kp = _case(n, 33)
    kp[..., 2] = 0.0
    op = InnerFitOP(bm, vp, n, iters_per_stage=5)
    out = op.fitting(init, kp).cpu().numpy()
    np.testing.assert_allclose(out[:, 0:3], init[:, 0:3], atol=1e-6)          # transl
    np.testing.assert_allclose(out[:, 72:75], init[:, 72:75], atol=1e-6)      # camera translation
    np.testing.assert_allclose(out[:, 3:6], init[:, 3:6], atol=2e-5)          # global_orient (6D <-> aa round trip)
    big = np.abs(init[:, 16:48]) > 0.2
    assert np.all(np.abs(out[:, 16:48])[big] < np.abs(init[:, 16:48])[big])
    op.close()
