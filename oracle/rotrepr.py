"""Restatement of cvae.ContinousRotReprDecoder and the parameter-vector conversions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/cvae.py:62-93 and
/root/reference/global_optimization.py:96-115.
"""
import torch
import torch.nn.functional as F

from . import tgm


def decode_6d(module_input: torch.Tensor) -> torch.Tensor:
    """cvae.py:62-72 -- view(-1,3,2), Gram-Schmidt, basis vectors stacked as COLUMNS."""
    a = module_input.reshape(-1, 3, 2)
    b1 = F.normalize(a[:, :, 0], dim=1)
    dot = torch.sum(b1 * a[:, :, 1], dim=1, keepdim=True)
    b2 = F.normalize(a[:, :, 1] - dot * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def matrot2aa(pose_matrot: torch.Tensor) -> torch.Tensor:
    """cvae.py:75-84 -- pad 3x3 to 3x4, tgm.rotation_matrix_to_angle_axis."""
    homogen = F.pad(pose_matrot.reshape(-1, 3, 3), [0, 1])
    return tgm.rotation_matrix_to_angle_axis(homogen).view(-1, 3).contiguous()


def aa2matrot(pose: torch.Tensor) -> torch.Tensor:
    """cvae.py:87-93."""
    return tgm.angle_axis_to_rotation_matrix(pose.reshape(-1, 3))[:, :3, :3].contiguous()


def convert_to_6D_rot(x_batch: torch.Tensor) -> torch.Tensor:
    """global_optimization.py:96-104: [N,75] -> [N,78]; 6D = first two COLUMNS of R,
    flattened row-major (R00,R01,R10,R11,R20,R21)."""
    xt = x_batch[:, :3]
    xr = x_batch[:, 3:6]
    xb = x_batch[:, 6:]
    xr_mat = aa2matrot(xr)
    xr_repr = xr_mat[:, :, :-1].reshape([-1, 6])
    return torch.cat([xt, xr_repr, xb], dim=-1)


def convert_to_3D_rot(x_batch: torch.Tensor) -> torch.Tensor:
    """global_optimization.py:107-115: [N,78] -> [N,75]."""
    xt = x_batch[:, :3]
    xr = x_batch[:, 3:9]
    xb = x_batch[:, 9:]
    xr_mat = decode_6d(xr)
    xr_aa = matrot2aa(xr_mat)
    return torch.cat([xt, xr_aa, xb], dim=-1)
