"""CPU tests of SURVEY.md §8f F2: mode 'dct' (global_optimization.py:232-246, :595-630) and the per-frame
smoother (optimization.py:155-238, :334-348).  Oracle vs the reference-generated goldens; the kernels' math
headers (compiled for the host, tests/cpu_harness) vs the oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import synth
from fdcap_amd.io import load_dct_base
from oracle.smoother import SmootherOracle
from tests import host_pipeline
from tests.host_pipeline import P, f32

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lib():
    lib = host_pipeline.build()
    lib.h_dct_joint_grad.restype = ctypes.c_double
    return lib


def test_oracle_smoother_reproduces_the_reference_run():
    g = np.load(os.path.join(GOLD, "ref_smoother.npz"))
    out = SmootherOracle(init_lr_h=float(g["lr"]), num_iter=int(g["num_iter"])).fitting_clip(g["body_in"]).numpy()
    # same torch ops in the same order as optimization.py: identical to the last bit on this torch build;
    # 1e-6 leaves room for another BLAS / vector width
    np.testing.assert_allclose(out, g["body_out"], rtol=0, atol=1e-6)


def test_host_smoother_math_matches_the_reference_run(lib):
    g = np.load(os.path.join(GOLD, "ref_smoother.npz"))
    x75 = f32(g["body_in"])
    n = x75.shape[0]
    x78 = np.zeros((n, 78), np.float32)
    lib.h_75_to_78(P(x75), n, P(x78))
    o78 = np.zeros_like(x78)
    lib.h_frame_smoother(P(x78), n, 50, ctypes.c_double(0.1), ctypes.c_float(1.0), ctypes.c_float(0.001), ctypes.c_float(5.0),
                         P(o78))
    o75 = np.zeros((n, 75), np.float32)
    lib.h_78_to_75(P(o78), n, P(o75))
    d = np.abs(o75 - g["body_out"])
    # measured: max 1.2e-6 (every element is an independent scalar problem; the host math follows torch's
    # Adam operation for operation, so even the L1 kinks are crossed at the same iterations)
    print("smoother host-vs-reference: max", d.max(), "q99", np.quantile(d, 0.99))
    assert d.max() < 1e-5


def _dct_problem(seed, W=2, T=60, C=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    D = load_dct_base(None, T, C)
    t = np.linspace(0, 1, W * T)[:, None]
    Jw = (rng.standard_normal((1, 69)) + 0.5 * np.sin(2 * np.pi * (t * rng.uniform(0.5, 2, (1, 69)) + rng.uniform(0, 1, (1, 69))))
          + 0.02 * rng.standard_normal((W * T, 69))).astype(np.float32)
    c0 = rng.standard_normal((W, 23, 3, C)).astype(np.float32)
    return D, Jw, c0


def _oracle_dct_loss(D, Jw, c, W, T):
    traj = Jw[:T * W].reshape(W, T, 23, 3)
    pred = torch.einsum("tc,kijc->ktij", D, c)
    err = (traj - pred) ** 2
    return torch.mean(torch.sum(err / (err + 1.0), dim=1))


def test_host_dct_fit_matches_torch_adam(lib):
    W, T, C, iters = 2, 60, 5, 400
    D, Jw, c0 = _dct_problem(3, W, T, C)
    c = torch.tensor(c0, requires_grad=True)
    opt = torch.optim.Adam([c], lr=0.005)
    hist = []
    for _ in range(iters):
        opt.zero_grad()
        l = _oracle_dct_loss(torch.tensor(D), torch.tensor(Jw), c, W, T)
        hist.append(float(l.detach()))
        (l * 10).backward()
        opt.step()
    coef = c0.reshape(W * 69, C).copy()
    m = np.zeros_like(coef)
    v = np.zeros_like(coef)
    obj = np.zeros((W * 69, iters), np.float32)
    for k in range(W):
        for ij in range(69):
            r = k * 69 + ij
            traj = f32(Jw[k * T:(k + 1) * T, ij])
            lib.h_dct_fit(P(traj), T, C, P(D), P(coef[r]), P(m[r]), P(v[r]), iters, 0, ctypes.c_double(0.005),
                          ctypes.c_float(10.0 / (69 * W)), P(obj[r]))
    # smooth objective (no kinks): the two Adam trajectories stay within fp32 rounding of each other
    np.testing.assert_allclose(coef.reshape(W, 23, 3, C), c.detach().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(obj.sum(0) / (69 * W), np.array(hist), rtol=2e-5, atol=1e-6)


def test_host_dct_joint_gradient_matches_autograd(lib):
    W, T, C = 2, 60, 5
    D, Jw, c0 = _dct_problem(5, W, T, C)
    n = W * T + 7                                         # 7 trailing frames outside every window: zero gradient
    Jw = np.concatenate([Jw, Jw[:7]], 0)
    J = torch.tensor(Jw, dtype=torch.float64, requires_grad=True)
    l = _oracle_dct_loss(torch.tensor(D, dtype=torch.float64), J, torch.tensor(c0, dtype=torch.float64), W, T)
    (l * 1e-4).backward()
    dJw = np.zeros((n, 69), np.float32)
    s = lib.h_dct_joint_grad(P(f32(Jw)), n, T, C, W, P(D), P(f32(c0.reshape(-1))), ctypes.c_float(1e-4 / (69 * W)), P(dJw))
    np.testing.assert_allclose(s / (69 * W), float(l), rtol=1e-6)
    g = J.grad.numpy()
    np.testing.assert_allclose(dJw, g, rtol=2e-5, atol=1e-6 * np.abs(g).max())
    assert np.all(dJw[W * T:] == 0)


def test_oracle_dct_mode_reproduces_the_reference_run():
    """oracle/fitting.py fitting_dct vs the reference's own 10000-iteration 'dct' run
    (tests/golden/make_golden.py --dct).  ~80 s."""
    from oracle.fitting import FittingOracle
    from oracle.smplx import SMPLXOracle
    from oracle.vposer import VPoserDecoder
    g = np.load(os.path.join(GOLD, "ref_dct_10000it.npz"))
    bm = synth.make_body_model(int(g["num_verts"]), seed=int(g["model_seed"]))
    vp = synth.make_vposer(seed=int(g["vposer_seed"]))
    orc = FittingOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp), g["scene"], g["vid"], list(g["camerapose"]), 300,
                        dct_mtx=g["dct_mtx"], c_dct_init=g["c_dct0"])
    body, scale, _ = orc.fitting_dct(torch.tensor(g["body_in"]))
    np.testing.assert_array_equal(orc.idx1, g["idx1"])
    log, ref = np.array(orc.loss_log), g["logd"]
    it = ref[:, 0].astype(int)
    first = it <= 9501                                     # c_dct phase + the no-op iteration 9500 + the first body step's forward
    np.testing.assert_allclose(log[it[first]], ref[first, 1:], rtol=5e-7, atol=4e-6)    # 6 printed decimals + fp32 rounding of values up to 186
    nxt = (it > 9501) & (it < 9506)
    np.testing.assert_allclose(log[it[nxt]], ref[nxt, 1:], rtol=5e-7, atol=4e-6)
    dc = np.abs(orc.c_dct.detach().numpy() - g["c_dct"])
    assert np.quantile(dc, 0.9) < 5e-6 and dc.max() < 2e-3, (np.quantile(dc, 0.9), dc.max())   # measured 5e-7 / 7e-4
    err = np.abs(body.numpy() - g["body_rec"])
    q50, q99 = np.quantile(err, [0.5, 0.99])
    # 499 Adam steps over an L1 data term (sign flips at kinks move an entry by +-lr): measured q50 2.5e-4, q99 3e-3, max 1.3e-2
    assert q50 < 1e-3 and q99 < 1e-2 and err.max() <= 2 * 0.005 * 499
    assert abs(float(scale) - float(g["scale"])) < 2e-3
