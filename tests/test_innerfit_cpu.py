"""CPU test of the per-frame inner fit's loss math: csrc/fdc_fit2d.h compiled for the host vs the
oracle's autograd (oracle/innerfit.py).  The objective itself is unpinned by the reference (it lives in SMPLify-X)."""
import ctypes

import numpy as np
import torch

import fdcap_amd  # noqa: F401
from tests import host_pipeline
from tests.host_pipeline import P, f32


def test_reprojection_and_prior_gradients_match_autograd():
    lib = host_pipeline.build()
    rng = np.random.Generator(np.random.PCG64(2))
    n = 5
    X = f32(rng.standard_normal((n, 78)) * 0.5)
    Jw = f32(rng.standard_normal((n, 23, 3)) * 0.4 + np.array([0.0, 0.0, 3.0]))
    kp = f32(np.concatenate([rng.uniform(100, 1100, (n, 23, 2)), rng.uniform(0.0, 1.0, (n, 23, 1))], -1))
    kp[0, 3, 2] = 0.0                                          # an undetected joint
    stage = f32([692, 692, 640, 360, 100, 1.3, 4.78, 5.0, 2.5])
    dX = np.zeros_like(X); dJw = np.zeros_like(Jw); losses = np.zeros(2)
    lib.h_fit2d_loss(P(stage), P(X), P(Jw), P(kp), n, P(dX), P(dJw), P(losses, ctypes.POINTER(ctypes.c_double)))
    x = torch.tensor(X, dtype=torch.float64, requires_grad=True)
    J = torch.tensor(Jw, dtype=torch.float64, requires_grad=True)
    k = torch.tensor(kp, dtype=torch.float64)
    fx, fy, cx, cy, rho, wd, wp, ws, wh = [float(v) for v in stage]
    uv = torch.stack([fx * J[..., 0] / J[..., 2] + cx, fy * J[..., 1] / J[..., 2] + cy], -1)
    r2 = (k[..., :2] - uv) ** 2
    data = wd ** 2 * torch.sum(k[..., 2:3] ** 2 * rho ** 2 * r2 / (r2 + rho ** 2))
    prior = wp ** 2 * torch.sum(x[:, 19:51] ** 2) + ws ** 2 * torch.sum(x[:, 9:19] ** 2) + wh ** 2 * torch.sum(x[:, 51:75] ** 2)
    (data + prior).backward()
    np.testing.assert_allclose(losses, [float(data), float(prior)], rtol=2e-6)
    np.testing.assert_allclose(dX, x.grad.numpy(), rtol=1e-6, atol=1e-9)
    g = J.grad.numpy()
    np.testing.assert_allclose(dJw, g, rtol=2e-5, atol=1e-6 * np.abs(g).max())
    assert np.all(dJw[0, 3] == 0)
