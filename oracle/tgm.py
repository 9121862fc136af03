"""Restatement of the two torchgeometry (0.1.2 API) functions the reference calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torchgeometry is not vendored in /root/reference
and not installed here; call sites: cvae.py:83 (rotation_matrix_to_angle_axis) and cvae.py:92
(angle_axis_to_rotation_matrix).  Algorithm: SURVEY.md Appendix A.1.  Parity unpinned against
the real package; self-consistency (aa -> R -> aa round trip, all four quaternion branches) is
tested in tests/test_oracle_units.py.
"""
import torch


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,4,4] (callers slice [:, :3, :3], cvae.py:92)."""
    eps = 1e-6
    theta2 = (angle_axis * angle_axis).sum(dim=1, keepdim=True)          # [N,1]
    theta = torch.sqrt(theta2)
    wxyz = angle_axis / (theta + eps)                                     # NOT exactly unit
    wx, wy, wz = wxyz[:, 0:1], wxyz[:, 1:2], wxyz[:, 2:3]
    c = torch.cos(theta)
    s = torch.sin(theta)
    one = 1.0
    r00 = c + wx * wx * (one - c)
    r10 = wz * s + wx * wy * (one - c)
    r20 = -wy * s + wx * wz * (one - c)
    r01 = wx * wy * (one - c) - wz * s
    r11 = c + wy * wy * (one - c)
    r21 = wx * s + wy * wz * (one - c)
    r02 = wy * s + wx * wz * (one - c)
    r12 = -wx * s + wy * wz * (one - c)
    r22 = c + wz * wz * (one - c)
    normal = torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)

    rx, ry, rz = angle_axis[:, 0:1], angle_axis[:, 1:2], angle_axis[:, 2:3]
    k1 = torch.ones_like(rx)
    taylor = torch.cat([k1, -rz, ry, rz, k1, -rx, -ry, rx, k1], dim=1).view(-1, 3, 3)

    mask = (theta2 > eps).view(-1, 1, 1)
    mask_pos = mask.type_as(theta2)
    mask_neg = (~mask).type_as(theta2)
    n = angle_axis.shape[0]
    out = torch.eye(4, dtype=angle_axis.dtype, device=angle_axis.device).view(1, 4, 4).repeat(n, 1, 1)
    out[..., :3, :3] = mask_pos * normal + mask_neg * taylor
    return out


def rotation_matrix_to_quaternion(rotation_matrix: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """[N,3,4] -> [N,4] (w,x,y,z).  The library transposes first; masks blend by multiply."""
    rt = rotation_matrix.transpose(1, 2)
    mask_d2 = rt[:, 2, 2] < eps
    mask_d0_d1 = rt[:, 0, 0] > rt[:, 1, 1]
    mask_d0_nd1 = rt[:, 0, 0] < -rt[:, 1, 1]

    t0 = 1 + rt[:, 0, 0] - rt[:, 1, 1] - rt[:, 2, 2]
    q0 = torch.stack([rt[:, 1, 2] - rt[:, 2, 1], t0, rt[:, 0, 1] + rt[:, 1, 0],
                      rt[:, 2, 0] + rt[:, 0, 2]], -1)
    t1 = 1 - rt[:, 0, 0] + rt[:, 1, 1] - rt[:, 2, 2]
    q1 = torch.stack([rt[:, 2, 0] - rt[:, 0, 2], rt[:, 0, 1] + rt[:, 1, 0], t1,
                      rt[:, 1, 2] + rt[:, 2, 1]], -1)
    t2 = 1 - rt[:, 0, 0] - rt[:, 1, 1] + rt[:, 2, 2]
    q2 = torch.stack([rt[:, 0, 1] - rt[:, 1, 0], rt[:, 2, 0] + rt[:, 0, 2],
                      rt[:, 1, 2] + rt[:, 2, 1], t2], -1)
    t3 = 1 + rt[:, 0, 0] + rt[:, 1, 1] + rt[:, 2, 2]
    q3 = torch.stack([t3, rt[:, 1, 2] - rt[:, 2, 1], rt[:, 2, 0] - rt[:, 0, 2],
                      rt[:, 0, 1] - rt[:, 1, 0]], -1)

    # the released source writes `1 - mask` on bool tensors (raises on torch>=1.2); users
    # patch it to `~mask` -- same truth table
    c0 = (mask_d2 & mask_d0_d1).view(-1, 1).type_as(q0)
    c1 = (mask_d2 & ~mask_d0_d1).view(-1, 1).type_as(q0)
    c2 = (~mask_d2 & mask_d0_nd1).view(-1, 1).type_as(q0)
    c3 = (~mask_d2 & ~mask_d0_nd1).view(-1, 1).type_as(q0)

    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 +
                       t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    q = q * 0.5
    return q


def quaternion_to_angle_axis(quaternion: torch.Tensor) -> torch.Tensor:
    q1 = quaternion[..., 1]
    q2 = quaternion[..., 2]
    q3 = quaternion[..., 3]
    sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3
    sin_theta = torch.sqrt(sin_squared_theta)
    cos_theta = quaternion[..., 0]
    two_theta = 2.0 * torch.where(cos_theta < 0.0,
                                  torch.atan2(-sin_theta, -cos_theta),
                                  torch.atan2(sin_theta, cos_theta))
    k_pos = two_theta / sin_theta
    k_neg = 2.0 * torch.ones_like(sin_theta)
    k = torch.where(sin_squared_theta > 0.0, k_pos, k_neg)
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(rotation_matrix: torch.Tensor) -> torch.Tensor:
    """[N,3,4] -> [N,3]."""
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(rotation_matrix))
