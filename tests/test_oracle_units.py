"""CPU unit tests of the oracle's small restatements: torchgeometry's two conversions (oracle/tgm.py, SURVEY Appendix A.1;
call sites /root/reference/cvae.py:83, :92) on every branch, and chamfer_python.py (oracle/chamfer.py) against the fixture
the reference's own file produced (tests/golden/ref_chamfer_python.npz, make_golden.py --chamfer)."""
import os

import numpy as np
import pytest
import torch

from oracle import tgm
from oracle.chamfer import NN_loss, distChamfer, nn_direct, pairwise_dist


def _rodrigues(aa):
    """Exact Rodrigues in fp64 (independent of the restatement under test)."""
    th = np.linalg.norm(aa, axis=1)
    out = np.zeros((aa.shape[0], 3, 3))
    for i, (r, t) in enumerate(zip(aa, th)):
        K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        if t < 1e-12:
            out[i] = np.eye(3) + K
        else:
            K = K / t
            out[i] = np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)
    return out


def _branch_of(R):
    """Which of the four quaternion candidates rotation_matrix_to_quaternion selects (it transposes first)."""
    rt = np.transpose(R, (0, 2, 1))
    d2 = rt[:, 2, 2] < 1e-6
    d01 = rt[:, 0, 0] > rt[:, 1, 1]
    d0n1 = rt[:, 0, 0] < -rt[:, 1, 1]
    return np.where(d2 & d01, 0, np.where(d2 & ~d01, 1, np.where(~d2 & d0n1, 2, 3)))


def _pad34(R):
    return torch.nn.functional.pad(torch.as_tensor(R), [0, 1])


def test_round_trip_covers_all_four_quaternion_branches():
    rng = np.random.default_rng(0)
    axis = rng.standard_normal((20000, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    aa = axis * rng.uniform(0.0, 3.1, (20000, 1))
    R = tgm.angle_axis_to_rotation_matrix(torch.tensor(aa))[:, :3, :3]
    np.testing.assert_allclose(R.numpy(), _rodrigues(aa), atol=3e-6)        # the +1e-6 in the axis normalisation
    br = _branch_of(R.numpy())
    share = np.bincount(br, minlength=4) / len(br)
    assert (share > 0.05).all(), share                                      # every candidate really selected
    back = tgm.rotation_matrix_to_angle_axis(_pad34(R)).numpy()
    for k in range(4):
        err = np.abs(back[br == k] - aa[br == k]).max()
        assert err < 5e-6, (k, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
def test_each_branch_on_an_exact_rotation(dtype, tol):
    """Rotations built by exact Rodrigues that land in each branch by construction: about x / y / z by 170 degrees
    (trace < 0, largest diagonal element decides) and a small generic one (branch 3)."""
    cases = {0: np.array([[np.deg2rad(170.0), 0.02, -0.01]]), 1: np.array([[0.02, np.deg2rad(170.0), 0.01]]),
             2: np.array([[0.01, -0.02, np.deg2rad(170.0)]]), 3: np.array([[0.2, -0.3, 0.1]])}
    for want, aa in cases.items():
        R = _rodrigues(aa)
        assert _branch_of(R)[0] == want
        got = tgm.rotation_matrix_to_angle_axis(_pad34(R).to(dtype)).double().numpy()
        np.testing.assert_allclose(got, aa, atol=tol)


def test_small_angle_limits():
    # theta^2 <= 1e-6: the first-order branch R = I + [aa]x
    aa = torch.tensor([[1e-4, -2e-4, 3e-4], [0.0, 0.0, 0.0]], dtype=torch.float64)
    R = tgm.angle_axis_to_rotation_matrix(aa)[:, :3, :3].numpy()
    np.testing.assert_array_equal(R[1], np.eye(3))
    np.testing.assert_allclose(R[0], np.eye(3) + np.array([[0, -3e-4, -2e-4], [3e-4, 0, -1e-4], [2e-4, 1e-4, 0]]), atol=0)
    # identity -> quaternion (1,0,0,0) -> s^2 = 0 -> k = 2 -> aa = 0 exactly (no 0/0 in the value)
    back = tgm.rotation_matrix_to_angle_axis(_pad34(np.eye(3)[None]))
    np.testing.assert_array_equal(back.numpy(), np.zeros((1, 3)))
    # ... while the gradient through the unselected k_pos branch is NaN there: the in-loop path avoids R -> aa
    Rg = torch.eye(3, dtype=torch.float64)[None].clone().requires_grad_(True)
    tgm.rotation_matrix_to_angle_axis(torch.nn.functional.pad(Rg, [0, 1])).sum().backward()
    assert torch.isnan(Rg.grad).any()


def test_angle_near_pi():
    rng = np.random.default_rng(1)
    axis = rng.standard_normal((500, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    aa = axis * (np.pi - rng.uniform(1e-4, 1e-2, (500, 1)))
    R = _rodrigues(aa)
    back = tgm.rotation_matrix_to_angle_axis(_pad34(R)).numpy()
    assert set(_branch_of(R)) <= {0, 1, 2}                                  # trace ~ -1: never the t3 candidate
    np.testing.assert_allclose(back, aa, atol=1e-8)
    # exactly pi about z: w = 0, cos_theta < 0.0 is False -> atan2(s, 0) = pi/2 -> angle pi
    Rz = np.diag([-1.0, -1.0, 1.0])[None]
    np.testing.assert_allclose(tgm.rotation_matrix_to_angle_axis(_pad34(Rz)).numpy(), [[0, 0, np.pi]], atol=1e-12)


# ---- chamfer_python.py (A18) ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def cham(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_chamfer_python.npz"))


def test_pairwise_dist_and_nn_loss_match_the_reference_file(cham):
    for tag in ("a", "b", "c"):
        x, y = torch.tensor(cham[f"{tag}_x"]), torch.tensor(cham[f"{tag}_y"])
        P = pairwise_dist(x, y)
        scale2 = float((x * x).sum(1).max() + (y * y).sum(1).max())
        tol = 4e-7 * scale2                                                 # mm kernels may round differently host to host
        if f"{tag}_P" in cham.files:
            np.testing.assert_allclose(P.numpy(), cham[f"{tag}_P"], rtol=0, atol=tol)
        else:
            for dim in (0, 1):
                np.testing.assert_allclose(P.min(dim=dim)[0].numpy(), cham[f"{tag}_Pmin{dim}"], rtol=0, atol=tol)
        np.testing.assert_allclose(float(NN_loss(x, y, dim=0)), float(cham[f"{tag}_nn0"]), rtol=0, atol=tol)
        np.testing.assert_allclose(float(NN_loss(x, y, dim=1)), float(cham[f"{tag}_nn1"]), rtol=0, atol=tol)
        # the expansion form against the direct-difference form the CUDA extension (and the HIP kernel) use
        d_xy, _ = nn_direct(x, y)
        np.testing.assert_allclose(P.min(dim=1)[0].numpy(), d_xy.numpy(), rtol=0, atol=4e-6 * scale2)


def test_distchamfer_return_order(cham):
    a, b = torch.tensor(cham["d_a"]), torch.tensor(cham["d_b"])
    r = distChamfer(a, b)
    scale2 = float((a * a).sum(-1).max() + (b * b).sum(-1).max())
    for k in range(2):
        np.testing.assert_allclose(r[k].numpy(), cham[f"d_ret{k}"], rtol=0, atol=4e-7 * scale2)
    for k in (2, 3):
        assert (r[k].numpy() == cham[f"d_ret{k}"]).mean() > 0.99
    # ret0 = per-y distances (y -> x), ret1 = per-x (x -> y): the opposite of chamferDist's (dist1, dist2)
    for bi in range(a.shape[0]):
        d_xy, i_xy = nn_direct(a[bi], b[bi])
        d_yx, i_yx = nn_direct(b[bi], a[bi])
        np.testing.assert_allclose(cham["d_ret1"][bi], d_xy.numpy(), rtol=0, atol=4e-6 * scale2)
        np.testing.assert_allclose(cham["d_ret0"][bi], d_yx.numpy(), rtol=0, atol=4e-6 * scale2)
        assert (cham["d_ret3"][bi] == i_xy.numpy()).mean() > 0.98 and (cham["d_ret2"][bi] == i_yx.numpy()).mean() > 0.98
